export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 200 python scripts/gpu_duo_profile.py --so wavernn_amd/csrc/libwrnn_profsplit.so --depth 4 --B 256 --T 600 --out gpurun_out/r03ag_duo_split.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-260
