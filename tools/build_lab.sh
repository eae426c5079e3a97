#!/bin/bash
# Lab builds of libibloc_hip.so with other compile-time constants (not the product library): tools/_lab/libibloc_<tag>.so,
# selected at run time with IBLOC_LIB=...   usage: tools/build_lab.sh <tag> <file.hip> "<-D flags>" [<file2.hip> ...]
set -e
tag=$1; shift
cd "$(dirname "$0")/../instance-based-loc_amd/csrc"
mkdir -p ../../tools/_lab
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I. -I../../include -Wall -Wno-unused-function -ffp-contract=on"
objs=$(ls *.o)
while [ $# -gt 0 ]; do
  f=$1; d=$2; shift 2
  /opt/rocm/bin/hipcc $FLAGS $d -c $f -o ../../tools/_lab/${f%.*}_$tag.o
  objs=$(echo $objs | sed "s/\b${f%.*}\.o\b//")
  objs="$objs ../../tools/_lab/${f%.*}_$tag.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_lab/libibloc_$tag.so $objs -lpthread -L/opt/rocm/lib -lrccl
echo built tools/_lab/libibloc_$tag.so
