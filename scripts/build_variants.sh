# builds tuning variants of the library into build/variants/ (they travel to the GPU box with the snapshot)
set -e
cd "$(dirname "$0")/../multigridcmt_amd/csrc"
rm -rf ../../build/variants; mkdir -p ../../build/variants
build() { name=$1; shift; make -s -j8 OUT=$PWD/../../build/variants/lib_$name.so OBJDIR=$PWD/../../build/variants/obj_$name EXTRA="$*"; echo built $name; }
build w1_minrows4_d91 -DMGCMT_FUSED_MIN_ROWS=4 -DMGCMT_FUSED_DEPTH9=1
build w1_minrows4 -DMGCMT_FUSED_MIN_ROWS=4
build w1_d2 -DMGCMT_FUSED_DEPTH=2
build w1_d94 -DMGCMT_FUSED_DEPTH9=4
build w1_d8 -DMGCMT_FUSED_DEPTH=8
