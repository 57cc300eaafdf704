# alternative builds of csrc/resstack.hip for tools/rs_probe.py (ablations of the convolution kernel): librs_<tag>.so under tools/rs_variants/
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/rs_variants
C=scl-deepfake-audio-detection_amd/csrc
for v in "nomfma:-DRS_NO_MFMA" "noepi:-DRS_NO_EPI" "noload:-DRS_NO_LOAD"; do
  tag=${v%%:*}; def=${v#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-gpu-rdc -shared $def $C/resstack.hip $C/api.hip -o tools/rs_variants/librs_$tag.so
done
ls -la tools/rs_variants
