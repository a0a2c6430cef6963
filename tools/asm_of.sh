#!/bin/bash
# device assembly of one source of the library:  tools/asm_of.sh train_dw.hip /tmp/train_dw.s [extra flags]
src=$1; out=$2; shift 2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I/root/repo/include -I/root/repo/st-nerf_amd/csrc "$@" -S --cuda-device-only /root/repo/st-nerf_amd/csrc/$src -o $out 2>&1 | grep -v hip-link
