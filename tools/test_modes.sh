# Runs the GPU parity tests under every kernel-selection / fallback switch the
# library still has.
for mode in "A=1" "SARA_HIP_GRAPH=0" "SARA_HIP_STREAMS=1" "SARA_HIP_BLUR=tile" \
            "SARA_HIP_FEATURES=tile" "SARA_HIP_SIDE_GRADIENT=0" \
            "SARA_HIP_MARCH_MIN_PIXELS=4194304" "SARA_HIP_XCD_MAP=0" \
            "SARA_HIP_STRIP_GROUP=0" "SARA_HIP_STRIP_GROUP=4" "SARA_HIP_STRIP_GROUP=1" \
            "SARA_HIP_OCTAVE_PIPELINE=0" "SARA_HIP_OCTAVE_PIPELINE=1" \
            "SARA_HIP_GRAPH=0 SARA_HIP_OCTAVE_PIPELINE=1" \
            "SARA_HIP_GRAD_TILE_PIXELS=0" "SARA_HIP_GRAD_TILE_PIXELS=100000000000" \
            "SARA_HIP_STREAMS=1 SARA_HIP_SIDE_GRADIENT=0" "SARA_HIP_LEVELS=0"; do
  echo "== $mode"
  env $mode python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_operators.py tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -1
done
# the two producers of the matcher's neighbour lists, forced for every size
for mode in "SARA_HIP_MATCH=mfma" "SARA_HIP_MATCH=exhaustive" "SARA_HIP_MATCH=mfma SARA_HIP_MATCH_PASSES=2"; do
  echo "== $mode"
  env $mode python -m pytest tests/test_gpu_matching.py tests/test_gpu_cpp_shim.py -m gpu -x -q 2>&1 | tail -1
done
