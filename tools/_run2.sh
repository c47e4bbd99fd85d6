export TMPDIR=/tmp; repo=$GRAFT_REPO_ROOT; cd /tmp
pass() { # name, counters...
  local name=$1; shift
  rm -rf /tmp/pm_$name
  timeout 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pm_$name -o kb -- python $repo/tools/kernel_bench.py --shape 3d --reps 3 --only "fwd C=" > /tmp/pm_$name.log 2>&1
  echo "pass $name rc $?"
  python $repo/tools/pmc_summary.py /tmp/pm_$name $repo/gpurun_out/aff_$name.csv
}
pass d SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_WAVES SQ_BUSY_CYCLES
pass c TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pass b TCP_PENDING_STALL_CYCLES_sum TCP_TAGRAM0_REQ_sum TCP_TA_TCP_STATE_READ_sum
cat $repo/gpurun_out/aff_d.csv $repo/gpurun_out/aff_c.csv $repo/gpurun_out/aff_b.csv | grep -E "affine_warp_fwd|k_sample_march<1|k_sample_march<4|grid_sample_fwd" | cut -c1-200
