#!/bin/bash
# A/B builds of the library for tools/ab_entropy.py: the product objects of dark_amd/_build/, with abi.cpp and entropy.cpp compiled again
# under extra flags.   usage: tools/build_variant.sh NAME "<flags for abi.cpp>" "<flags for entropy.cpp>"   ->  tools/_ab/NAME.so
set -e
cd "$(dirname "$0")/.."
NAME=$1; ABI_FLAGS=$2; ENT_FLAGS=$3
B=dark_amd/_build; O=tools/_ab/obj_$NAME; mkdir -p $O
COMMON="-O3 -std=c++17 -fPIC -Wall -Wno-unused-result --offload-arch=gfx950 -march=x86-64-v3"
/opt/rocm/bin/hipcc $COMMON $ABI_FLAGS -c dark_amd/csrc/abi.cpp -o $O/abi.o
/opt/rocm/bin/hipcc $COMMON -mllvm -inline-threshold=20000 $ENT_FLAGS -c dark_amd/csrc/entropy.cpp -o $O/entropy.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_ab/$NAME.so $O/abi.o $B/context.o $O/entropy.o $B/bbb.o $B/radix_sort.o $B/suffix_array.o $B/bwt.o $B/dc.o
rm -rf $O
echo tools/_ab/$NAME.so
