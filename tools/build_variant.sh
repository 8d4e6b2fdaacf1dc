#!/bin/bash
# A/B builds of ONE translation unit with extra -D flags: tools/build_variant.sh NAME FILE.hip "-DX=1 -DY=2"
# -> shacira_amd/lib/variants/NAME.so (select with SHACIRA_HIP_LIB=...). Flags and the other objects come from the Makefile.
set -e
make -s -j4 -C "$(dirname "$0")/../shacira_amd/csrc" variant NAME="$1" FILE="$2" EXTRA="$3"
echo built shacira_amd/lib/variants/$1.so
