# after the builder threads' opt-out: the data path and the bench step with the shipped defaults (launch thread: upload stream; builders: direct copies)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r6_upload_stream_final.txt; : > $o
timeout 900 python -m pytest tests/test_augment_gpu.py tests/test_pack_gpu.py tests/test_dropout_gpu.py tests/test_model_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -3 >> $o
for v in 1 0 1; do echo "SCL_UPLOAD_STREAM=$v (builder threads copy on their own stream either way)" >> $o; SCL_UPLOAD_STREAM=$v PROBE_PARTS=12 timeout 900 python tools/data_path_probe.py 2>&1 | grep "PACKS=\|sampler fast      builder threads 1" >> $o; done
for i in 1 2 3; do python bench.py 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench default: %.2f ms/step %.0f utt/s frac %.3f' % (d['ms_per_step'], d['value'], d['roofline']['frac']))" >> $o; done
cat $o
