set -e
cd "$(dirname "$0")/../multigridcmt_amd/csrc"
rm -rf ../../build/variants ../../variants; mkdir -p ../../build/variants ../../variants
build() { name=$1; shift; make -s -j8 OUT=$PWD/../../variants/lib_$name.so OBJDIR=$PWD/../../build/variants/obj_$name "$@"; echo built $name; }
# round 2: prefetch depth of the 9-point (Galerkin-level) fused passes, 1 (adopted) against 3
build depth9_3 EXTRA=-DMGCMT_FUSED_DEPTH9=3
# diagnostic build of the lexicographic wave pipeline (per-block timing words; scripts/lex_debug.py)
build lexdebug EXTRA=-DMGCMT_LEXWAVE_DEBUG
# chained lexicographic sweeps: records asked for 6 rows ahead (default 4; must stay below the old values' distance)
build chainrec6 EXTRA=-DMGCMT_LEX_CHAIN_REC=6
