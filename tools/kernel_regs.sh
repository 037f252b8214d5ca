#!/bin/bash
# Compact register / LDS / occupancy table of the kernels in one .hip file (hipcc remarks):
#   tools/kernel_regs.sh medicalseg_amd/csrc/msk_conv_wbf.hip [name-filter]
f=$1; pat=${2:-.}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Imedicalseg_amd/csrc -Wno-unused-value -Wno-comment \
  -Rpass-analysis=kernel-resource-usage -c "$f" -o /dev/null 2>&1 |
  awk '/Function Name:/{n=$NF} / VGPRs:/{v=$(NF-1)} /AGPRs:/{a=$(NF-1)} /VGPRs Spill/{s=$(NF-1)} /Occupancy/{o=$(NF-1)} /LDS Size/{l=$(NF-2); print n, "vgpr="v, "agpr="a, "spill="s, "occ="o, "lds="l}' |
  sed 's/ \[-Rpass[^ ]*//g' | c++filt | grep -E "$pat"
