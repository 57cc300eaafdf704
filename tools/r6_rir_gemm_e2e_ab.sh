cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r6_call11.txt; : > $o
for i in 1 2; do for v in 0 1; do
  echo "SCL_RIR_GEMM=$v" >> $o
  SCL_RIR_GEMM=$v PROBE_PARTS=2 timeout 900 python tools/data_path_probe.py 2>&1 | grep "PACKS=" >> $o
done; done
cat $o
