export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05b
python $R/tools/ab/graph_solver_ab.py --modes eager,graph 2>&1 | grep -v "control point\|amdgpu.ids\|UserWarning\|Consider\|print(" 
python $R/tools/ab/graph_solver_ab.py --modes graph --margin 1.0 2>&1 | grep -v "control point\|amdgpu.ids\|UserWarning\|Consider\|print(" | head -8
for m in eager graph; do
  rm -rf /tmp/rp_$m
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$m -o $m -- python $R/tools/ab/graph_solver_ab.py --modes $m --steps 20 > $R/gpurun_out/r05b/${m}_rocprof.log 2>&1
  cp $(find /tmp/rp_$m -name "${m}_kernel_stats.csv" | head -1) $R/gpurun_out/r05b/${m}_kernel_stats.csv
done
