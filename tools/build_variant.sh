#!/bin/bash
# usage: tools/build_variant.sh <name> [extra -D flags...]   builds both libraries of the current tree into
# nimble-aligner_amd/libv/<name>/ (select with NIMBLE_LIB_DIR; ships to the GPU box with gpurun)
NAME=$1; shift
cd "$(dirname "$0")/../nimble-aligner_amd" || exit 1
make LIBDIR=libv/$NAME OBJDIR=/tmp/nimble_build_$NAME HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wextra -Wno-unused-parameter $*" 2>&1 | grep -E "error|Error"
ls libv/$NAME
