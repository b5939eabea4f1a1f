#!/bin/bash
# Build variants/libv2p_<name>.so from the sources of a git ref (default HEAD) for A/B runs with tools/variants.sh.
# usage: tools/mkvariant.sh <name> [ref | WORK = the working tree] ; extra per-file flags through V2P_FLAGS_PHYSICS_LL etc.
set -e
NAME=$1; REF=${2:-HEAD}; ROOT=$(cd "$(dirname "$0")/.." && pwd); TMP=$(mktemp -d)
if [ "$REF" = WORK ]; then (cd $ROOT && tar -c --exclude='*.o' --exclude='*.so' --exclude=__pycache__ vid2player3d_amd include) | tar -x -C $TMP
else git -C $ROOT archive $REF vid2player3d_amd include | tar -x -C $TMP; fi
(cd $TMP && python -c "
from vid2player3d_amd import build
build.build(force=True, lib_out='$ROOT/variants/libv2p_$NAME.so')" > /dev/null)
rm -rf $TMP; ls -la $ROOT/variants/libv2p_$NAME.so
