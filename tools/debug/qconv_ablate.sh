#!/bin/bash
# Where does qconv_kernel's time go?  tools/qconv_bench.py under the kernel's diagnostic switches (CAT_Q_ABLATE bits, results become wrong):
# 1 = filter-stream loads hit one cached block, 4 = no staging / barrier after the first chunk, 16 = no output stores, 32 = no statistics
# epilogue, 64 = one MFMA step per chunk (what is left: prologue + staging + epilogue), 128 = no staging loads for the first chunk
export CAT_LIB=diag      # the ablation switches exist only in the diagnostic build: python -m cat_amd._build --diag
for m in ${MODES:-0 16 64 80 208 212}; do echo "== CAT_Q_ABLATE=$m"; CAT_Q_ABLATE=$m python tools/qconv_bench.py "$@" 2>/dev/null | tail -n +2; done
