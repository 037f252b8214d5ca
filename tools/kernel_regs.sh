#!/bin/bash
# Compact register / LDS / occupancy table of the kernels in one .hip file (hipcc remarks):
#   tools/kernel_regs.sh medicalseg_amd/csrc/msk_conv_wbf.hip [name-filter]
f=$1; pat=${2:-.}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Imedicalseg_amd/csrc -Wno-unused-value -Wno-comment \
  -Rpass-analysis=kernel-resource-usage -c "$f" -o /dev/null 2>&1 | sed 's/ \[-Rpass[^ ]*\]//' |
  awk '/Function Name:/{n=$NF} / VGPRs:/{v=$NF} /AGPRs:/{a=$NF} /VGPRs Spill/{s=$NF} /Occupancy/{o=$NF} /LDS Size/{l=$NF; print n, "vgpr="v, "agpr="a, "spill="s, "occ="o, "lds="l}' |
  c++filt | grep -E "$pat"
