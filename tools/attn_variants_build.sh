# alternative builds of csrc/attention.hip for tools/attn_variants_probe.py (ablations / variants of the fused forward): libattn_<tag>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/attn_variants
C=scl-deepfake-audio-detection_amd/csrc
for v in "base:-DATT_BASE" "noqpre:-DATT_NO_QPRE" "noexp:-DATT_ABL_NOEXP" "noqk:-DATT_ABL_NOQK" "nopv:-DATT_ABL_NOPV" "nostage:-DATT_ABL_NOSTAGE" "nomfma:-DATT_ABL_NOQK -DATT_ABL_NOPV" "valuonly:-DATT_ABL_NOQK -DATT_ABL_NOPV -DATT_ABL_NOSTAGE"; do
  tag=${v%%:*}; def=${v#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-gpu-rdc -shared $def $C/attention.hip $C/api.hip -o tools/attn_variants/libattn_$tag.so &
  if (( $(jobs -r | wc -l) >= 6 )); then wait -n; fi
done
wait
ls tools/attn_variants
