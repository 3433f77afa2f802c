for p in 3 2 1; do W2XC_BF16_PIPE=split python tools/layer_bench.py --precision $p --steps 10 2>&1 | tail -1; done
