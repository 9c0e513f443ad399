# builds tuning variants of the library into build/variants/ (they travel to the GPU box with the snapshot)
set -e
cd "$(dirname "$0")/../multigridcmt_amd/csrc"
rm -rf ../../build/variants; mkdir -p ../../build/variants
build() { name=$1; shift; make -s -j8 OUT=$PWD/../../build/variants/lib_$name.so OBJDIR=$PWD/../../build/variants/obj_$name EXTRA="$*"; echo built $name; }
build nts -DMGCMT_FUSED_NT_STORE=1
build ntsf -DMGCMT_FUSED_NT_STORE=1 -DMGCMT_FUSED_NT_F=1
build d2 -DMGCMT_FUSED_DEPTH=2
build d2nts -DMGCMT_FUSED_DEPTH=2 -DMGCMT_FUSED_NT_STORE=1
