#!/bin/bash
# Development aid: rebuild only the named objects of a sampler library (e.g. "specs_4_64 specp_4_64 mchap_hip") and link it from
# them plus the objects that are already there (stale, but link-compatible while no shared struct changed).  A release build is
# `make -j8` in mchap_amd/csrc.  Usage: bash tools/quick_build.sh [OBJ=obj_phases EXTRA=-DMCHAP_PHASES OUT=libmchap_hip_phases.so] name...
cd "$(dirname "$0")/../mchap_amd/csrc" || exit 1
vars=()
names=()
for a in "$@"; do case "$a" in *=*) vars+=("$a");; *) names+=("$a");; esac; done
obj=obj
out=libmchap_hip.so
for v in "${vars[@]}"; do case "$v" in OBJ=*) obj=${v#OBJ=};; OUT=*) out=${v#OUT=};; esac; done
targets=()
for n in "${names[@]}"; do targets+=("$obj/$n.o"); done
make -j8 "${vars[@]}" "${targets[@]}" || exit 1
make -t "${vars[@]}" "$out" > /dev/null || exit 1   # every other object counts as up to date
rm -f "$out"
make "${vars[@]}" "$out" | tail -1 | cut -c1-80
