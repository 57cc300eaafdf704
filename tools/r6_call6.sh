cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r6_call6.txt; : > $o
timeout 1500 python tools/data_path_probe.py 2>&1 | grep -v "amdgpu.ids\|Scores saved\|vocoders" > gpurun_out/r6_pack_builder.txt
cat gpurun_out/r6_pack_builder.txt >> $o
echo "== Toeplitz f32-MFMA form of the FIR / RIR convolutions vs fir_kernel" >> $o
timeout 600 python tools/fir_toeplitz_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_fir_toeplitz_probe.txt
cat gpurun_out/r6_fir_toeplitz_probe.txt >> $o
echo "== attention probe" >> $o
python tools/attn_probe.py 64 32 2>&1 | grep -v amdgpu >> $o
echo "== full GPU suite" >> $o
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -6 >> $o
cat $o
