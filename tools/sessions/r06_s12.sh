#!/bin/bash
# round 6, GPU session 12: chunk-length sweep of the two sub-voxel north-star kernels under rocprofv3 (the pair alone)
set -u
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd $repo
mkdir -p gpurun_out/r06zc
for rep in 1 2; do
for fz in 0 3 5 6 8; do
  ADVCHAIN_FWD_MARCH_ZC=$fz bash tools/profile_ns_pair.sh r06zc_f$fz > /dev/null 2>&1
  echo "fwd zc=$fz  $(grep -E 'k_sample_march' gpurun_out/r06zc_f$fz/ns_pair_summary.txt | head -1)"
done
for bz in 0 4 6 10 12 16; do
  ADVCHAIN_MARCH_ZC=$bz bash tools/profile_ns_pair.sh r06zc_b$bz > /dev/null 2>&1
  echo "bwd zc=$bz  $(grep -E 'k_adjoint_march' gpurun_out/r06zc_b$bz/ns_pair_summary.txt | head -1)"
done
done | tee gpurun_out/r06zc/zc_sweep.txt
