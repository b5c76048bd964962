export TMPDIR=/tmp; mkdir -p gpurun_out
for so in libwavernn_amd libwrnn_fcglb libwrnn_fcglb_xa libwrnn_fcglb_xas; do
  echo "== $so"
  timeout 200 python scripts/gpu_perf_probe.py --so wavernn_amd/csrc/$so.so --T 1000 --B 192,256,512 --variants d3,d4,d8 --out gpurun_out/r03x_probe_$so.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-110
done
