# builds tuning variants of the library into build/variants/ (they travel to the GPU box with the snapshot)
set -e
cd "$(dirname "$0")/../multigridcmt_amd/csrc"
rm -rf ../../build/variants; mkdir -p ../../build/variants
build() { name=$1; shift; make -s -j8 OUT=$PWD/../../build/variants/lib_$name.so OBJDIR=$PWD/../../build/variants/obj_$name EXTRA="$*"; echo built $name; }
build d9_4 -DMGCMT_FUSED_DEPTH9=4
build d9_1 -DMGCMT_FUSED_DEPTH9=1
build w2 -DMGCMT_FUSED_WAVES=2
