# register / LDS / spill summary of every kernel of ccsx_kernels.hip (no GPU needed): bash tools/kres.sh [extra hipcc flags]
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Wno-unused-value -fno-slp-vectorize -falign-loops=64 \
  -Iinclude -Iccs_amd/csrc "$@" -Rpass-analysis=kernel-resource-usage -c ccs_amd/csrc/ccsx_kernels.hip -o /tmp/kres.o 2>&1 |
  grep -E "Function Name|VGPRs:|SGPRs Spill|VGPRs Spill|ScratchSize|LDS Size|Occupancy" | sed 's/.*remark: //' | paste - - - - - - - | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g'
