cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r6_call14.txt; : > $o
timeout 900 python -m pytest tests/test_pack_gpu.py tests/test_augment_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -3 >> $o
for i in 1 2; do
  PROBE_PARTS=12 timeout 1200 python tools/data_path_probe.py 2>&1 | grep "PACKS=\|sampler fast" >> $o
done
cat $o
