#!/bin/bash
# Here (no GPU): a second library for same-box A/B timing - the current objects with ONE source file taken from a git revision.
#   tools/build_ab_lib.sh <rev> <file under tuatara_amd/csrc>   ->  tuatara_amd/lib/libtuatara_hip_ab.so   (use: TUATARA_LIB=.../libtuatara_hip_ab.so)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
rev=$1; f=$2
mkdir -p /tmp/ab_build
git -C $R show $rev:tuatara_amd/csrc/$f > $R/tuatara_amd/csrc/_ab_$f
trap "rm -f $R/tuatara_amd/csrc/_ab_$f" EXIT
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c $R/tuatara_amd/csrc/_ab_$f -o /tmp/ab_build/$f.o
objs=$(ls $R/build/obj/*.o | grep -v "/$f.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/tuatara_amd/lib/libtuatara_hip_ab.so $objs /tmp/ab_build/$f.o -ldl -L/opt/rocm/lib -lrccl
ls -la $R/tuatara_amd/lib/
