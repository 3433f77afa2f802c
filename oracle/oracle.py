"""ctypes binding of the CPU checkers -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (waifu2x-converter-cpp_amd/) never does.

Two checkers:
  * ``Oracle``  -- oracle/_build/libw2xc_oracle.so, the C restatement (w2xc_oracle.c) of
    Model::filter / convertWithModels (/root/reference/src/modelHandler.cpp:26-159,
    src/convertRoutine.cpp:21-169).
  * ``RefBuild`` -- oracle/_ref/libw2xc_ref.so, the reference's own two source files compiled
    against the OpenCV shim (prebuilt in the build container; /root/reference is not needed at
    run time).
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "_build", "libw2xc_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libw2xc_ref.so")


def build(quiet=True):
    """Compile the checkers (gcc/g++).  oracle/_ref is only rebuilt where /root/reference exists."""
    subprocess.run(["make", "-C", _HERE] + (["-s"] if quiet else []), check=True)


class _Layer(C.Structure):
    _fields_ = [("nin", C.c_int), ("nout", C.c_int),
                ("weight", C.POINTER(C.c_float)), ("bias", C.POINTER(C.c_double))]


def load_model_json(path):
    """Parse the reference JSON schema (export_model_nocuda.lua:12-19) the way the reference does
    (modelHandler.cpp:74-115): JSON number -> double (picojson strtod == Python float) -> float
    for weights, double for biases.  Returns [(nin, nout, W[o,i,3,3] float32, bias[o] float64)]."""
    with open(path) as f:
        objs = json.load(f)
    layers = []
    for o in objs:
        nin, nout = int(o["nInputPlane"]), int(o["nOutputPlane"])
        if int(o["kW"]) != int(o["kH"]):
            raise ValueError("kernel in model is not square")   # modelHandler.hpp:52-58
        w = np.asarray(o["weight"], dtype=np.float64).astype(np.float32)
        b = np.asarray(o["bias"], dtype=np.float64)
        assert w.shape == (nout, nin, int(o["kH"]), int(o["kW"])) and b.shape == (nout,)
        layers.append((nin, nout, np.ascontiguousarray(w), np.ascontiguousarray(b)))
    return layers


class Oracle:
    def __init__(self, layers):
        if not os.path.exists(ORACLE_SO):
            build()
        self.lib = C.CDLL(ORACLE_SO)
        self.layers = layers
        self._arr = (_Layer * len(layers))()
        for k, (nin, nout, w, b) in enumerate(layers):
            assert w.dtype == np.float32 and b.dtype == np.float64
            self._arr[k] = _Layer(nin, nout, w.ctypes.data_as(C.POINTER(C.c_float)),
                                  b.ctypes.data_as(C.POINTER(C.c_double)))
        L = self.lib
        L.w2xc_oracle_filter.restype = C.c_int
        L.w2xc_oracle_filter.argtypes = [C.POINTER(_Layer), C.c_int, C.c_void_p, C.c_int, C.c_int,
                                         C.c_void_p, C.c_int]
        L.w2xc_oracle_convert.restype = C.c_int
        L.w2xc_oracle_convert.argtypes = [C.POINTER(_Layer), C.c_int, C.c_void_p, C.c_size_t, C.c_int,
                                          C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                          C.c_int]
        L.w2xc_oracle_convert_f64.restype = C.c_int
        L.w2xc_oracle_convert_f64.argtypes = [C.POINTER(_Layer), C.c_int, C.c_void_p, C.c_size_t, C.c_int,
                                              C.c_int, C.c_void_p, C.c_int]

    @classmethod
    def from_json(cls, path):
        return cls(load_model_json(path))

    def filter(self, layer, planes, njob=4):
        """Model::filter: planes [nin,h,w] float32 -> [nout,h,w]; None on plane-count mismatch."""
        planes = np.ascontiguousarray(planes, dtype=np.float32)
        n, h, w = planes.shape
        out = np.empty((self.layers[layer][1], h, w), np.float32)
        rc = self.lib.w2xc_oracle_filter(C.byref(self._arr[layer]), n, planes.ctypes.data, w, h,
                                         out.ctypes.data, njob)
        return out if rc == 0 else None

    def convert(self, plane, block_splitting=True, block=(512, 512), njob=4):
        """convertWithModels on one H x W float32 plane."""
        plane = np.ascontiguousarray(plane, dtype=np.float32)
        h, w = plane.shape
        out = np.empty((h, w), np.float32)
        rc = self.lib.w2xc_oracle_convert(self._arr, len(self.layers), plane.ctypes.data, w, w, h,
                                          out.ctypes.data, w, int(block_splitting), block[0], block[1], njob)
        if rc != 0:
            raise RuntimeError("oracle convert failed rc=%d" % rc)
        return out

    def convert_f64(self, plane, njob=8):
        plane = np.ascontiguousarray(plane, dtype=np.float32)
        h, w = plane.shape
        out = np.empty((h, w), np.float64)
        rc = self.lib.w2xc_oracle_convert_f64(self._arr, len(self.layers), plane.ctypes.data, w, w, h,
                                              out.ctypes.data, njob)
        if rc != 0:
            raise RuntimeError("oracle convert_f64 failed rc=%d" % rc)
        return out


class RefBuild:
    """The reference's own modelHandler.cpp/convertRoutine.cpp (+ picojson) over the OpenCV shim."""

    def __init__(self, json_path):
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO + " (built by oracle/Makefile where /root/reference exists)")
        self.lib = C.CDLL(REF_SO)
        L = self.lib
        L.w2xc_ref_load.restype = C.c_void_p
        L.w2xc_ref_load.argtypes = [C.c_char_p]
        L.w2xc_ref_free.argtypes = [C.c_void_p]
        for f in (L.w2xc_ref_nlayers,):
            f.restype = C.c_int
            f.argtypes = [C.c_void_p]
        for f in (L.w2xc_ref_nin, L.w2xc_ref_nout):
            f.restype = C.c_int
            f.argtypes = [C.c_void_p, C.c_int]
        L.w2xc_ref_convert.restype = C.c_int
        L.w2xc_ref_convert.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                       C.c_int, C.c_int, C.c_int]
        L.w2xc_ref_filter.restype = C.c_int
        L.w2xc_ref_filter.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                      C.c_void_p, C.c_int]
        self.h = L.w2xc_ref_load(json_path.encode())
        if not self.h:
            raise RuntimeError("reference generateModelFromJSON returned false for " + json_path)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.w2xc_ref_free(self.h)
            self.h = None

    @property
    def nlayers(self):
        return self.lib.w2xc_ref_nlayers(self.h)

    def planes(self, l):
        return self.lib.w2xc_ref_nin(self.h, l), self.lib.w2xc_ref_nout(self.h, l)

    def convert(self, plane, block_splitting=True, block=(512, 512), njob=4):
        plane = np.ascontiguousarray(plane, dtype=np.float32)
        h, w = plane.shape
        out = np.empty((h, w), np.float32)
        rc = self.lib.w2xc_ref_convert(self.h, plane.ctypes.data, w, h, out.ctypes.data,
                                       int(block_splitting), njob, block[0], block[1])
        if rc != 0:
            raise RuntimeError("reference convertWithModels failed")
        return out

    def filter(self, layer, planes, njob=4):
        planes = np.ascontiguousarray(planes, dtype=np.float32)
        n, h, w = planes.shape
        out = np.empty((self.planes(layer)[1], h, w), np.float32)
        rc = self.lib.w2xc_ref_filter(self.h, layer, n, planes.ctypes.data, w, h, out.ctypes.data, njob)
        return out if rc == 0 else None


# ---- N2: colour front/back end and U/V resize of the CLI scale loop (w2xc_oracle_color.c) ---------------
def _color_lib():
    if not os.path.exists(ORACLE_SO):
        build()
    lib = C.CDLL(ORACLE_SO)
    if not hasattr(lib, "w2xc_oracle_u8_to_yuv"):
        build()
        lib = C.CDLL(ORACLE_SO)
    return lib


def u8_to_yuv(img):
    """main.cpp:75-76 (+split): h x w x 3 uint8 -> (Y, U, V) float32 planes."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, _ = img.shape
    y, u, v = (np.empty((h, w), np.float32) for _ in range(3))
    _color_lib().w2xc_oracle_u8_to_yuv(C.c_void_p(img.ctypes.data), C.c_size_t(w * 3), w, h, C.c_void_p(y.ctypes.data),
                                      C.c_void_p(u.ctypes.data), C.c_void_p(v.ctypes.data))
    return y, u, v


def yuv_to_u8(y, u, v):
    """main.cpp:171-172 (+merge): float32 planes -> h x w x 3 uint8."""
    y, u, v = (np.ascontiguousarray(a, dtype=np.float32) for a in (y, u, v))
    h, w = y.shape
    out = np.empty((h, w, 3), np.uint8)
    _color_lib().w2xc_oracle_yuv_to_u8(C.c_void_p(y.ctypes.data), C.c_void_p(u.ctypes.data), C.c_void_p(v.ctypes.data), w, h,
                                      C.c_void_p(out.ctypes.data), C.c_size_t(w * 3))
    return out


def resize2x_cubic(plane):
    plane = np.ascontiguousarray(plane, dtype=np.float32)
    h, w = plane.shape
    out = np.empty((2 * h, 2 * w), np.float32)
    _color_lib().w2xc_oracle_resize2x_cubic(C.c_void_p(plane.ctypes.data), w, h, C.c_void_p(out.ctypes.data))
    return out


def resize2x_nearest(plane):
    plane = np.ascontiguousarray(plane, dtype=np.float32)
    h, w = plane.shape
    out = np.empty((2 * h, 2 * w), np.float32)
    _color_lib().w2xc_oracle_resize2x_nearest(C.c_void_p(plane.ctypes.data), w, h, C.c_void_p(out.ctypes.data))
    return out


def scale2x_image_u8(oracle, img, iterations=1):
    """The CLI's scale phase on a uint8 image (main.cpp:74-76,126-156,171-172): the image stays float YUV
    between 2x iterations; only the final convertTo clips."""
    y, u, v = u8_to_yuv(img)
    for _ in range(iterations):
        y = oracle.convert(resize2x_nearest(y))
        u, v = resize2x_cubic(u), resize2x_cubic(v)
    return yuv_to_u8(y, u, v)


def resize_linear(plane, dw, dh):
    """main.cpp:158-167 on one plane: cv::resize(..., INTER_LINEAR)."""
    plane = np.ascontiguousarray(plane, dtype=np.float32)
    h, w = plane.shape
    out = np.empty((dh, dw), np.float32)
    _color_lib().w2xc_oracle_resize_linear(C.c_void_p(plane.ctypes.data), w, h, C.c_void_p(out.ctypes.data), dw, dh)
    return out


def process_image_u8(img, noise_oracle=None, scale_oracle=None, iterations=0, shrink_ratio=0.0):
    """noise (main.cpp:83-98), `iterations` 2x steps (:126-156), optional INTER_LINEAR shrink (:158-167)."""
    y, u, v = u8_to_yuv(img)
    if noise_oracle is not None:
        y = noise_oracle.convert(y)
    for _ in range(iterations):
        y = scale_oracle.convert(resize2x_nearest(y))
        u, v = resize2x_cubic(u), resize2x_cubic(v)
    if shrink_ratio:
        h, w = y.shape
        dw, dh = int(float(w * shrink_ratio)), int(float(h * shrink_ratio))
        y, u, v = (resize_linear(p, dw, dh) for p in (y, u, v))
    return yuv_to_u8(y, u, v)
