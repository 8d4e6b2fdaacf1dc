#!/bin/bash
# A/B builds of ONE translation unit with extra -D flags: tools/build_variant.sh NAME FILE.hip "-DX=1 -DY=2"
# -> shacira_amd/lib/variants/NAME.so (select with SHACIRA_HIP_LIB=...). The other objects come from the normal build.
set -e
cd "$(dirname "$0")/../shacira_amd/csrc"
make -s -j4
NAME=$1; FILE=$2; FLAGS=$3; REPLACES=${4:-$2}   # 4th arg: the source whose object this build replaces
mkdir -p ../lib/variants
OBJ=../lib/variants/$NAME.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -munsafe-fp-atomics \
    -fvisibility=hidden $FLAGS -c $FILE -o $OBJ
OTHERS=$(ls ../lib/obj/*.o | grep -v "/${REPLACES%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/$NAME.so $OBJ $OTHERS
rm -f $OBJ
echo built ../lib/variants/$NAME.so
