# builds tuning variants of the library into build/variants/ (they travel to the GPU box with the snapshot)
set -e
cd "$(dirname "$0")/../multigridcmt_amd/csrc"
rm -rf ../../build/variants; mkdir -p ../../build/variants
build() { name=$1; shift; make -s -j8 OUT=$PWD/../../build/variants/lib_$name.so OBJDIR=$PWD/../../build/variants/obj_$name "$@"; echo built $name; }
build fma FUSED_FLAGS=-ffp-contract=fast
build fma_d2 FUSED_FLAGS="-ffp-contract=fast" EXTRA=-DMGCMT_FUSED_DEPTH=2
build d8 EXTRA=-DMGCMT_FUSED_DEPTH=8
