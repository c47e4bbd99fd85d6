python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "affine" 2>&1 | tail -2
for rot in 0 15; do for v in "" "ADVCHAIN_NO_AFFINE_ZG=1"; do echo "rot $rot $v"
KB_ROT=$rot env $v python tools/kernel_bench.py --shape 3d --only "affine_warp fwd C=4" 2>/dev/null | grep affine
done; done
