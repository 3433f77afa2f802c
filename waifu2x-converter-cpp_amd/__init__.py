"""waifu2x-converter-cpp_amd -- host-side mirror of the reference's hot-path interface over the
C ABI of lib/libw2xc_hip.so (include/w2xc_hip.h).

The directory name is not a Python identifier; load it with ``__graft_entry__.load_package()``
(registers it as module ``w2xc_amd``).

Names follow /root/reference/src/modelHandler.hpp and convertRoutine.hpp:

    models = []
    modelUtility.generateModelFromJSON("models/scale2.0x_model.json", models)   # modelHandler.cpp:170
    modelUtility.getInstance().setNumberOfJobs(4)                               # :199
    out = Mat()
    ok = convertWithModels(Mat(y_plane), out, models)                           # convertRoutine.cpp:21
    ok = models[0].filter(inputPlanes, outputPlanes)                            # modelHandler.cpp:26

``Mat`` is a minimal stand-in for a CV_32FC1 ``cv::Mat`` (a float32 numpy view, may be a strided
ROI).  All arithmetic happens in the HIP library; there is no CPU fallback -- if the shared
library is missing the import fails, and without a GPU every compute call returns ``False`` /
raises ``W2xcError``.  PyTorch is only used by callers for device memory and streams.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libw2xc_hip.so")

OK, ERR_IO, ERR_JSON, ERR_ARG, ERR_PLANES, ERR_HIP, ERR_UNSUPPORTED, ERR_NOMEM = 0, -1, -2, -3, -4, -5, -6, -7
PRECISION_FP32, PRECISION_BF16, PRECISION_BF16X2, PRECISION_BF16X3, PRECISION_FP16X2 = 0, 1, 2, 3, 4
KERNEL_AUTO, KERNEL_DIRECT, KERNEL_MFMA, KERNEL_WINOGRAD, KERNEL_WINOGRAD32, KERNEL_WINOGRAD4 = 0, 1, 2, 3, 4, 5
FUSION_AUTO, FUSION_OFF, FUSION_ON, FUSION_FIRST, FUSION_LAST, FUSION_GATHER_LAUNCH, FUSION_PROG = 0, 1, 2, 3, 4, 5, 6


class W2xcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("w2xc error %d: %s" % (code, msg))
        self.code = code


class Opts(C.Structure):
    """struct w2xc_opts (include/w2xc_hip.h)."""
    _fields_ = [("struct_size", C.c_int), ("precision", C.c_int), ("kernel", C.c_int), ("device", C.c_int),
                ("device_mask", C.c_uint), ("band_rows", C.c_int), ("workspace_mb", C.c_int),
                ("profile", C.c_int), ("verbose", C.c_int), ("filter_resident", C.c_int), ("fusion", C.c_int),
                ("host_units", C.c_int), ("host_chunk_kb", C.c_int), ("host_numa", C.c_int)]


class RowPlan(C.Structure):
    """struct w2xc_row_plan (include/w2xc_hip.h): the band geometry of one row-range conversion."""
    _fields_ = [("struct_size", C.c_int), ("n_layers", C.c_int), ("halo_rows_per_layer", C.c_int), ("band_rows", C.c_int),
                ("n_bands", C.c_int), ("fused_first", C.c_int), ("fused_last", C.c_int), ("workspace_bytes", C.c_ulonglong * 2)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError("HIP extension %s is missing -- build it with __graft_entry__.build() "
                          "(make -C waifu2x-converter-cpp_amd/csrc). There is no CPU fallback." % LIB_PATH)
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.7.  Callers use torch for device memory and
    # streams, so let torch's copy load first: libw2xc_hip.so then binds to the same (single) HIP runtime
    # and device pointers / streams can be shared.  Without torch the system ROCm runtime is used.
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except Exception:   # torch is plumbing, not a requirement
            pass
    lib = C.CDLL(LIB_PATH)
    vp, ci, cs, fp = C.c_void_p, C.c_int, C.c_size_t, C.c_void_p
    sig = {
        "w2xc_opts_init": (None, [C.POINTER(Opts)]),
        "w2xc_opts_init_sized": (None, [C.POINTER(Opts), cs]),
        "w2xc_set_default_opts": (ci, [C.POINTER(Opts)]),
        "w2xc_plan_rows": (ci, [vp, ci, ci, ci, ci, ci, ci, C.POINTER(Opts), C.POINTER(RowPlan)]),
        "w2xc_plan_region": (ci, [C.POINTER(RowPlan), ci, ci, ci, ci, C.POINTER(ci), C.POINTER(ci)]),
        "w2xc_model_load_json": (ci, [C.c_char_p, C.POINTER(vp)]),
        "w2xc_model_from_arrays": (ci, [ci, C.POINTER(ci), C.POINTER(ci), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]),
        "w2xc_model_free": (None, [vp]),
        "w2xc_model_trim": (ci, [vp]),
        "w2xc_model_layers": (ci, [vp]),
        "w2xc_model_nin": (ci, [vp, ci]),
        "w2xc_model_nout": (ci, [vp, ci]),
        "w2xc_model_get_layer": (ci, [vp, ci, fp, fp]),
        "w2xc_set_jobs": (ci, [ci]),
        "w2xc_get_jobs": (ci, []),
        "w2xc_set_block_size": (ci, [ci, ci]),
        "w2xc_set_block_size_exp2": (ci, [ci]),
        "w2xc_get_block_size": (None, [C.POINTER(ci), C.POINTER(ci)]),
        "w2xc_convert_plane": (ci, [vp, fp, cs, ci, ci, fp, cs, ci, C.POINTER(Opts)]),
        "w2xc_convert_plane_device": (ci, [vp, fp, cs, ci, ci, fp, cs, vp, C.POINTER(Opts)]),
        "w2xc_convert_planes_device": (ci, [vp, ci, fp, cs, cs, ci, ci, fp, cs, cs, vp, C.POINTER(Opts)]),
        "w2xc_process_image_u8_ex_device": (ci, [vp, vp, fp, cs, ci, ci, fp, cs, ci, C.c_double, vp, C.POINTER(Opts)]),
        "w2xc_process_image_u8_ex": (ci, [vp, vp, fp, cs, ci, ci, fp, cs, ci, C.c_double, C.POINTER(Opts)]),
        "w2xc_process_image_u8_device": (ci, [vp, vp, fp, cs, ci, ci, fp, cs, ci, vp, C.POINTER(Opts)]),
        "w2xc_process_image_u8": (ci, [vp, vp, fp, cs, ci, ci, fp, cs, ci, C.POINTER(Opts)]),
        "w2xc_scale2x_image_u8_device": (ci, [vp, fp, cs, ci, ci, fp, cs, ci, vp, C.POINTER(Opts)]),
        "w2xc_scale2x_image_u8": (ci, [vp, fp, cs, ci, ci, fp, cs, ci, C.POINTER(Opts)]),
        "w2xc_resize2x_cubic_device": (ci, [fp, ci, ci, fp, vp]),
        "w2xc_u8_to_yuv_device": (ci, [fp, cs, ci, ci, fp, fp, fp, vp]),
        "w2xc_yuv_to_u8_device": (ci, [fp, fp, fp, ci, ci, fp, cs, vp]),
        "w2xc_convert_plane_nn2x": (ci, [vp, fp, cs, ci, ci, fp, cs, C.POINTER(Opts)]),
        "w2xc_convert_plane_nn2x_device": (ci, [vp, fp, cs, ci, ci, fp, cs, vp, C.POINTER(Opts)]),
        "w2xc_convert_plane_rows": (ci, [vp, fp, cs, ci, ci, ci, ci, ci, ci, ci, fp, cs, C.POINTER(Opts)]),
        "w2xc_layer_filter_device": (ci, [vp, ci, ci, fp, C.c_longlong, C.c_longlong, C.c_longlong, ci, ci, fp, C.c_longlong,
                                          C.c_longlong, C.c_longlong, vp, C.POINTER(Opts)]),
        "w2xc_convert_rows_device": (ci, [vp, fp, cs, ci, ci, ci, ci, ci, ci, fp, cs, vp, C.POINTER(Opts)]),
        "w2xc_layer_filter": (ci, [vp, ci, ci, C.POINTER(fp), cs, ci, ci, C.POINTER(fp), cs, C.POINTER(Opts)]),
        "w2xc_profile_read": (ci, [vp, ci, C.POINTER(C.c_float), C.POINTER(ci), ci]),
        "w2xc_profile_reset": (None, [vp, ci]),
        "w2xc_layer_kernel_name": (C.c_char_p, [vp, ci, C.POINTER(Opts)]),
        "w2xc_device_count": (ci, []),
        "w2xc_last_error": (C.c_char_p, []),
        "w2xc_version": (C.c_char_p, []),
    }
    for name, (res, args) in sig.items():
        f = getattr(lib, name)   # AttributeError if the ABI lost a symbol
        f.restype = res
        f.argtypes = args
    return lib, tuple(sig)


_lib, ABI_SYMBOLS = _load()


def lib():
    return _lib


def last_error():
    return (_lib.w2xc_last_error() or b"").decode(errors="replace")


def device_count():
    return _lib.w2xc_device_count()


def make_opts(**kw):
    o = Opts()
    _lib.w2xc_opts_init(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise TypeError("unknown w2xc_opts field %r" % k)
        setattr(o, k, v)
    return o


# ---------------------------------------------------------------------------------------------
class Mat:
    """Stand-in for a CV_32FC1 cv::Mat: wraps a 2-D float32 array whose rows are contiguous
    (row stride arbitrary, like a cv::Mat ROI with its ``step``)."""

    def __init__(self, array=None):
        self.array = None
        if array is not None:
            a = np.asarray(array)
            if a.dtype != np.float32 or a.ndim != 2:
                a = np.ascontiguousarray(a, dtype=np.float32)
                if a.ndim != 2:
                    raise ValueError("Mat wants a 2-D plane")
            if a.strides[1] != 4:
                a = np.ascontiguousarray(a)
            self.array = a

    @property
    def rows(self):
        return 0 if self.array is None else self.array.shape[0]

    @property
    def cols(self):
        return 0 if self.array is None else self.array.shape[1]

    @property
    def step(self):
        return 0 if self.array is None else self.array.strides[0]

    def size(self):
        return (self.cols, self.rows)   # cv::Size(width, height)

    def create(self, rows, cols):
        if self.array is None or self.array.shape != (rows, cols):
            self.array = np.empty((rows, cols), np.float32)

    def copyTo(self, other):
        other.create(self.rows, self.cols)
        other.array[...] = self.array


class _ModelSet:
    """Owns one w2xc_model* (== the reference's std::vector<std::unique_ptr<Model>>)."""

    def __init__(self, handle):
        self.handle = C.c_void_p(handle)

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h and _lib is not None:
            _lib.w2xc_model_free(h)

    @classmethod
    def from_json(cls, path):
        h = C.c_void_p()
        rc = _lib.w2xc_model_load_json(os.fsencode(path), C.byref(h))
        if rc != OK:
            raise W2xcError(rc, last_error())
        return cls(h.value)

    @classmethod
    def from_layers(cls, layers):
        """layers: [(nin, nout, W[o,i,3,3] float32, bias[o] float64)]"""
        n = len(layers)
        nin = (C.c_int * n)(*[l[0] for l in layers])
        nout = (C.c_int * n)(*[l[1] for l in layers])
        ws = [np.ascontiguousarray(l[2], dtype=np.float32) for l in layers]
        bs = [np.ascontiguousarray(l[3], dtype=np.float64) for l in layers]
        for (ni, no, _, _), w, b in zip(layers, ws, bs):
            if w.shape != (no, ni, 3, 3) or b.shape != (no,):
                raise ValueError("layer arrays have the wrong shape")
        wp = (C.c_void_p * n)(*[w.ctypes.data for w in ws])
        bp = (C.c_void_p * n)(*[b.ctypes.data for b in bs])
        h = C.c_void_p()
        rc = _lib.w2xc_model_from_arrays(n, nin, nout, wp, bp, C.byref(h))
        if rc != OK:
            raise W2xcError(rc, last_error())
        return cls(h.value)

    def trim(self):
        """release the buffers that grew with the largest plane so far (w2xc_model_trim); weights stay resident"""
        rc = _lib.w2xc_model_trim(self.handle)
        if rc != OK:
            raise W2xcError(rc, last_error())

    @property
    def n_layers(self):
        return _lib.w2xc_model_layers(self.handle)

    def planes(self, l):
        return _lib.w2xc_model_nin(self.handle, l), _lib.w2xc_model_nout(self.handle, l)

    def layer_arrays(self, l):
        nin, nout = self.planes(l)
        w = np.empty((nout, nin, 3, 3), np.float32)
        b = np.empty((nout,), np.float64)
        rc = _lib.w2xc_model_get_layer(self.handle, l, w.ctypes.data, b.ctypes.data)
        if rc != OK:
            raise W2xcError(rc, last_error())
        return nin, nout, w, b

    # -- functional forms used by tests / bench ------------------------------------------------
    def convert(self, plane, block_splitting=True, opts=None):
        """convertWithModels on a host float32 plane -> new float32 plane (raises on failure)."""
        src = Mat(plane)
        out = np.empty((src.rows, src.cols), np.float32)
        rc = _lib.w2xc_convert_plane(self.handle, src.array.ctypes.data, src.step, src.cols, src.rows,
                                     out.ctypes.data, out.strides[0], int(block_splitting),
                                     C.byref(opts) if opts is not None else None)
        if rc != OK:
            raise W2xcError(rc, last_error())
        return out

    def convert_planes_device(self, n_in, d_in, in_plane_stride_bytes, in_stride_bytes, w, h, d_out,
                              out_plane_stride_bytes, out_stride_bytes, stream=0, opts=None):
        """Multi-plane wrapper (pad n / all layers / crop) on planar device planes; writes every plane of the
        last layer (w2xc_convert_planes_device)."""
        rc = _lib.w2xc_convert_planes_device(self.handle, n_in, C.c_void_p(d_in), in_plane_stride_bytes, in_stride_bytes, w, h,
                                             C.c_void_p(d_out), out_plane_stride_bytes, out_stride_bytes, C.c_void_p(stream),
                                             C.byref(opts) if opts is not None else None)
        if rc != OK:
            raise W2xcError(rc, last_error())

    def scale2x_image_u8(self, img, iterations=1, opts=None):
        """The CLI's scale phase on an h x w x 3 uint8 image (main.cpp:74-76,126-156,171-172) -> (h<<it) x (w<<it) x 3."""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w, ch = img.shape
        if ch != 3:
            raise ValueError("want h x w x 3 uint8")
        out = np.empty((h << iterations, w << iterations, 3), np.uint8)
        rc = _lib.w2xc_scale2x_image_u8(self.handle, img.ctypes.data, img.strides[0], w, h, out.ctypes.data, out.strides[0],
                                        iterations, C.byref(opts) if opts is not None else None)
        if rc != OK:
            raise W2xcError(rc, last_error())
        return out

    def scale2x_image_u8_device(self, d_in, in_stride_bytes, w, h, d_out, out_stride_bytes, iterations=1, stream=0, opts=None):
        rc = _lib.w2xc_scale2x_image_u8_device(self.handle, C.c_void_p(d_in), in_stride_bytes, w, h, C.c_void_p(d_out),
                                               out_stride_bytes, iterations, C.c_void_p(stream),
                                               C.byref(opts) if opts is not None else None)
        if rc != OK:
            raise W2xcError(rc, last_error())

    def convert_nn2x(self, plane, opts=None):
        """cv::resize(INTER_NEAREST, 2x) + convertWithModels (main.cpp:132-148) in one call: h x w -> 2h x 2w."""
        src = Mat(plane)
        out = np.empty((2 * src.rows, 2 * src.cols), np.float32)
        rc = _lib.w2xc_convert_plane_nn2x(self.handle, src.array.ctypes.data, src.step, src.cols, src.rows,
                                          out.ctypes.data, out.strides[0], C.byref(opts) if opts is not None else None)
        if rc != OK:
            raise W2xcError(rc, last_error())
        return out

    def convert_nn2x_device(self, d_in, in_stride_bytes, w, h, d_out, out_stride_bytes, stream=0, opts=None):
        rc = _lib.w2xc_convert_plane_nn2x_device(self.handle, C.c_void_p(d_in), in_stride_bytes, w, h, C.c_void_p(d_out),
                                                 out_stride_bytes, C.c_void_p(stream), C.byref(opts) if opts is not None else None)
        if rc != OK:
            raise W2xcError(rc, last_error())

    def convert_device(self, d_in, in_stride_bytes, w, h, d_out, out_stride_bytes, stream=0, opts=None):
        """Device-pointer form (ints: hipDeviceptr / hipStream_t).  Asynchronous on `stream`."""
        rc = _lib.w2xc_convert_plane_device(self.handle, C.c_void_p(d_in), in_stride_bytes, w, h,
                                            C.c_void_p(d_out), out_stride_bytes, C.c_void_p(stream),
                                            C.byref(opts) if opts is not None else None)
        if rc != OK:
            raise W2xcError(rc, last_error())

    def convert_rows_device(self, d_view, view_stride_bytes, view_h, view_y0, w, plane_h, row_begin, row_end,
                            d_out, out_stride_bytes, stream=0, opts=None):
        """Row-band form (one shard of a plane): see w2xc_convert_rows_device in include/w2xc_hip.h."""
        rc = _lib.w2xc_convert_rows_device(self.handle, C.c_void_p(d_view), view_stride_bytes, view_h, view_y0, w,
                                           plane_h, row_begin, row_end, C.c_void_p(d_out), out_stride_bytes,
                                           C.c_void_p(stream), C.byref(opts) if opts is not None else None)
        if rc != OK:
            raise W2xcError(rc, last_error())

    def convert_rows(self, view, view_y0, plane_h, row_begin, row_end, nn2x=0, opts=None, out=None):
        """One unit of the tile farm from HOST memory (w2xc_convert_plane_rows): `view` holds source rows
        [view_y0, view_y0 + len(view)) of a plane_h-row source plane; returns output rows [row_begin, row_end) (in
        output coordinates: of the 2x plane when nn2x = 1)."""
        src = Mat(view)
        w = src.cols
        if out is None:
            out = np.empty((row_end - row_begin, w << nn2x), np.float32)
        rc = _lib.w2xc_convert_plane_rows(self.handle, src.array.ctypes.data, src.step, view_y0, src.rows, w, plane_h, nn2x,
                                          row_begin, row_end, out.ctypes.data, out.strides[0],
                                          C.byref(opts) if opts is not None else None)
        if rc != OK:
            raise W2xcError(rc, last_error())
        return out

    def filter_device(self, layer, n_in, d_in, in_strides, w, h, d_out, out_strides, stream=0, opts=None):
        """Model::filter on device data; *_strides = (plane, row, pixel) element strides in floats (w2xc_layer_filter_device)."""
        rc = _lib.w2xc_layer_filter_device(self.handle, layer, n_in, C.c_void_p(d_in), in_strides[0], in_strides[1], in_strides[2], w, h,
                                           C.c_void_p(d_out), out_strides[0], out_strides[1], out_strides[2], C.c_void_p(stream),
                                           C.byref(opts) if opts is not None else None)
        if rc != OK:
            raise W2xcError(rc, last_error())

    def filter(self, layer, planes, opts=None):
        """Model::filter: [nin,h,w] -> [nout,h,w]; raises W2xcError(ERR_PLANES) on a plane-count mismatch."""
        planes = [Mat(p).array for p in planes]
        nout = self.planes(layer)[1]
        h, w = planes[0].shape
        strides = {p.strides[0] for p in planes}
        if len(strides) != 1:
            planes = [np.ascontiguousarray(p) for p in planes]
        out = np.empty((nout, h, w), np.float32)
        ip = (C.c_void_p * len(planes))(*[p.ctypes.data for p in planes])
        op = (C.c_void_p * nout)(*[out[o].ctypes.data for o in range(nout)])
        rc = _lib.w2xc_layer_filter(self.handle, layer, len(planes), ip, planes[0].strides[0], w, h, op,
                                    out.strides[1], C.byref(opts) if opts is not None else None)
        if rc != OK:
            raise W2xcError(rc, last_error())
        return out

    def plan_rows(self, w, plane_h, row_begin=0, row_end=None, view_y0=0, view_h=None, opts=None):
        """The band geometry a row-range conversion would run with (w2xc_plan_rows): no device needed."""
        row_end = plane_h if row_end is None else row_end
        view_h = plane_h - view_y0 if view_h is None else view_h
        p = RowPlan()
        p.struct_size = C.sizeof(RowPlan)
        rc = _lib.w2xc_plan_rows(self.handle, w, view_y0, view_h, plane_h, row_begin, row_end,
                                 C.byref(opts) if opts is not None else None, C.byref(p))
        if rc != OK:
            raise W2xcError(rc, last_error())
        return p

    @staticmethod
    def plan_region(plan, plane_h, layer, y0, y1):
        """plane rows [top, bottom) layer `layer` (1-based) computes for the band [y0, y1) under `plan` (w2xc_plan_region)"""
        t, b = C.c_int(), C.c_int()
        rc = _lib.w2xc_plan_region(C.byref(plan), plane_h, layer, y0, y1, C.byref(t), C.byref(b))
        if rc != OK:
            raise W2xcError(rc, last_error())
        return t.value, b.value

    def kernel_name(self, layer, opts=None):
        return _lib.w2xc_layer_kernel_name(self.handle, layer, C.byref(opts) if opts is not None else None).decode()

    def profile_read(self, device=-1):
        n = self.n_layers
        ms = (C.c_float * n)()
        cnt = (C.c_int * n)()
        rc = _lib.w2xc_profile_read(self.handle, device, ms, cnt, n)
        if rc != OK:
            raise W2xcError(rc, last_error())
        return list(ms), list(cnt)

    def profile_reset(self, device=-1):
        _lib.w2xc_profile_reset(self.handle, device)


class Model:
    """w2xc::Model -- ONE conv layer (modelHandler.hpp:24-90)."""

    def __init__(self, model_set, index):
        self._set, self._index = model_set, index

    def getNInputPlanes(self):
        return self._set.planes(self._index)[0]

    def getNOutputPlanes(self):
        return self._set.planes(self._index)[1]

    def filter(self, inputPlanes, outputPlanes):
        """bool filter(std::vector<cv::Mat>& in, std::vector<cv::Mat>& out) -- modelHandler.cpp:26-72."""
        try:
            out = self._set.filter(self._index, [m.array if isinstance(m, Mat) else m for m in inputPlanes])
        except W2xcError as e:
            sys.stderr.write(str(e) + "\n")
            return False
        del outputPlanes[:]
        outputPlanes.extend(Mat(out[o]) for o in range(out.shape[0]))
        return True

    def printWeightMatrix(self):   # modelHandler.cpp:229-235
        _, _, w, _ = self._set.layer_arrays(self._index)
        for k in w.reshape(-1, 3, 3):
            print(k)

    def printBiases(self):         # modelHandler.cpp:237-242
        for b in self._set.layer_arrays(self._index)[3]:
            print(b)


class modelUtility:
    """w2xc::modelUtility (modelHandler.hpp:92-113): JSON loader + nJob / block-size singleton."""
    _instance = None

    @staticmethod
    def generateModelFromJSON(fileName, models):
        try:
            s = _ModelSet.from_json(fileName)
        except W2xcError as e:
            sys.stderr.write(str(e) + "\n")
            return False
        models.extend(Model(s, i) for i in range(s.n_layers))
        return True

    @staticmethod
    def getInstance():
        if modelUtility._instance is None:
            modelUtility._instance = modelUtility()
        return modelUtility._instance

    def setNumberOfJobs(self, n):
        return _lib.w2xc_set_jobs(int(n)) == OK

    def getNumberOfJobs(self):
        return _lib.w2xc_get_jobs()

    def setBlockSize(self, size):
        return _lib.w2xc_set_block_size(int(size[0]), int(size[1])) == OK

    def setBlockSizeExp2Square(self, exp):
        return _lib.w2xc_set_block_size_exp2(int(exp)) == OK

    def getBlockSize(self):
        w, h = C.c_int(), C.c_int()
        _lib.w2xc_get_block_size(C.byref(w), C.byref(h))
        return (w.value, h.value)


def _set_of(models):
    """The w2xc_model* behind a list of Model layers.  A list that is exactly one loaded file in
    order reuses its handle (weights stay resident on the GPUs); any other selection / order of
    layers gets its own container, as the reference allows any vector of Models."""
    if not models:
        raise W2xcError(ERR_ARG, "empty model list")
    s = models[0]._set
    if all(m._set is s for m in models) and [m._index for m in models] == list(range(s.n_layers)):
        return s
    return _ModelSet.from_layers([m._set.layer_arrays(m._index) for m in models])


def convertWithModels(inputPlane, outputPlane, models, blockSplitting=True, opts=None):
    """bool w2xc::convertWithModels(cv::Mat& in, cv::Mat& out, std::vector<std::unique_ptr<Model>>&,
    bool blockSplitting = true) -- convertRoutine.cpp:21-51.  `outputPlane` (a Mat) is
    (re)allocated to the input size like cv::Mat::copyTo does (:46)."""
    try:
        s = _set_of(models)
        src = inputPlane if isinstance(inputPlane, Mat) else Mat(inputPlane)
        res = s.convert(src.array, blockSplitting, opts)
    except W2xcError as e:
        sys.stderr.write(str(e) + "\n")
        return False
    Mat(res).copyTo(outputPlane)
    return True


# ---- sharding one plane into independent row bands (multi-GPU: no exchange, host-side gather) ----
def shard_rows(plane_h, n_parts, part):
    """Output rows [begin, end) of shard `part` of `n_parts` (contiguous, sizes differ by <= 1 row).
    Same split as the in-process multi-device path of w2xc_convert_plane (csrc/w2xc_host_pipeline.cpp)."""
    return (plane_h * part) // n_parts, (plane_h * (part + 1)) // n_parts


def shard_view(plane_h, row_begin, row_end, n_layers):
    """Input rows [y0, y1) a shard needs: its rows plus an n_layers halo, clipped to the plane
    (the 2*nModel overlap of the reference's block split, convertRoutine.cpp:100-131).  This is the MINIMUM the row entry points accept;
    a shard that is to stitch BIT-identically with the whole-plane call passes the wide halo -- shard_view(h, ra, rb, 4 * n_layers) -- which the
    default F(4x4) mid-layer kernel needs for its banding-invariant geometry (on the minimum view W2XC_KERNEL_AUTO is refused with
    ERR_ARG; name a kernel -- KERNEL_WINOGRAD32 is banding-invariant there -- to use it)."""
    return max(0, row_begin - n_layers), min(plane_h, row_end + n_layers)


def process_image_u8(img, noise=None, scale=None, iterations=0, opts=None, shrink_ratio=0.0):
    """The CLI's processing modes on an h x w x 3 uint8 image (main.cpp -m noise | scale | noise_scale):
    `noise` / `scale` are _ModelSet objects (either may be None); shrink_ratio = the final INTER_LINEAR shrink
    of main.cpp:158-167 (0 = none)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, _ = img.shape
    fh, fw = h << iterations, w << iterations
    if shrink_ratio:
        fw, fh = int(float(fw * shrink_ratio)), int(float(fh * shrink_ratio))
    out = np.empty((fh, fw, 3), np.uint8)
    rc = _lib.w2xc_process_image_u8_ex(noise.handle if noise else None, scale.handle if scale else None, img.ctypes.data,
                                       img.strides[0], w, h, out.ctypes.data, out.strides[0], iterations, float(shrink_ratio),
                                       C.byref(opts) if opts is not None else None)
    if rc != OK:
        raise W2xcError(rc, last_error())
    return out
