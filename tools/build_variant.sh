#!/bin/bash
# usage: tools/build_variant.sh <name> [<git rev> | WORK] [extra hipcc flags ...]
# Builds libmaro_amd.so of the given revision's maro_amd/csrc (WORK: the working tree) into variants/<name>/ — an A/B build that
# travels to the GPU box with the snapshot (MARO_AMD_LIB=variants/<name>/libmaro_amd.so selects it; *.so is git-ignored).
set -e
name=$1; rev=${2:-WORK}; shift; shift || true
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/variants/$name; mkdir -p $out
src=$(mktemp -d)
if [ "$rev" = WORK ]; then cp -r $root/maro_amd/csrc/*.h $root/maro_amd/csrc/*.hip $src/; mkdir -p $src/../../include; else
  (cd $root && git archive $rev maro_amd/csrc include) | tar -x -C $src; mv $src/maro_amd/csrc/* $src/ 2>/dev/null || true; fi
# the sources include "../../include/maro_amd.h" relative to maro_amd/csrc: recreate that layout
lay=$(mktemp -d); mkdir -p $lay/maro_amd/csrc $lay/include
cp $src/*.h $src/*.hip $lay/maro_amd/csrc/ 2>/dev/null
if [ "$rev" = WORK ]; then cp $root/include/*.h $lay/include/; else cp $src/include/*.h $lay/include/; fi
(cd $lay/maro_amd/csrc && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value "$@" -o $out/libmaro_amd.so cim_engine.hip cb_engine.hip 2>/dev/null)
rm -rf $src $lay
ls -la $out/libmaro_amd.so
