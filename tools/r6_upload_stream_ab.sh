# bench step and pack builder with host -> device copies on the compute stream (0) vs on an upload stream (1, default)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r6_upload_stream_ab.txt; : > $o
timeout 900 python -m pytest tests/test_augment_gpu.py tests/test_pack_gpu.py tests/test_dropout_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -3 >> $o
bash tools/ab_env.sh "SCL_UPLOAD_STREAM=0" "SCL_UPLOAD_STREAM=1" 3 >> $o 2>&1
bash tools/ab_env.sh "SCL_UPLOAD_STREAM=0" "SCL_UPLOAD_STREAM=1" 2 --batch 11 --steps 20 >> $o 2>&1
for v in 0 1 0 1; do echo "SCL_UPLOAD_STREAM=$v" >> $o; SCL_UPLOAD_STREAM=$v PROBE_PARTS=12 timeout 900 python tools/data_path_probe.py 2>&1 | grep "PACKS=\|sampler fast      builder threads 1" >> $o; done
python3 tools/phase_gaps.py events 2>/dev/null | grep '^{' | tail -1 >> $o
cat $o
