# deep-ring 128 x 128 kernel vs the two-stage one at pack-sized M, kernel time from the launches' own dispatch stamps; needs experiment builds:
#   for s in 3 5; do SCL_BUILD_TAG=s$s SCL_BUILD_DEFINES="-DSCL_EXPERIMENTS -DSCL_DEEP_STAGES=$s" python scl-deepfake-audio-detection_amd/build.py; done
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r6_small_m_probe.txt; : > $o
for s in 5 3; do
  echo "== experiment build, ring of $s stages" >> $o
  SCL_LIB_PATH=$GRAFT_REPO_ROOT/scl-deepfake-audio-detection_amd/build_s$s/libscl_hip.so timeout 600 python tools/small_m_probe.py 2189 2>&1 | grep -v amdgpu.ids >> $o
done
echo "== shipped library (the deep columns repeat the two-stage kernel)" >> $o
timeout 600 python tools/small_m_probe.py 2189 6368 2>&1 | grep -v amdgpu.ids >> $o
cat $o
