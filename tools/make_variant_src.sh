#!/bin/bash
# A patched copy of the kernel sources for an A/B that must not touch the tree (the tree's source digest keys profiles/pmc_traffic.json):
#   tools/make_variant_src.sh <name> tools/experiments/<a>.patch [<b>.patch ...]   ->  build_variants/src_<name>/   (then: LG_VARIANT_SRC=$PWD/build_variants/src_<name> tools/build_variant.sh <name>)
set -e
NAME=$1; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
D="$ROOT/build_variants/src_$NAME"
rm -rf "$D"; mkdir -p "$ROOT/build_variants"; cp -r "$ROOT/lightglue_amd/csrc" "$D"
for p in "$@"; do (cd "$D" && patch -p3 --no-backup-if-mismatch < "$ROOT/$p"); done
echo "$D"
