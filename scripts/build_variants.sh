set -e
cd "$(dirname "$0")/../multigridcmt_amd/csrc"
rm -rf ../../build/variants; mkdir -p ../../build/variants
build() { name=$1; shift; make -s -j8 OUT=$PWD/../../build/variants/lib_$name.so OBJDIR=$PWD/../../build/variants/obj_$name "$@"; echo built $name; }
build d1 EXTRA=-DMGCMT_FUSED_DEPTH=1
build d9_3 EXTRA=-DMGCMT_FUSED_DEPTH9=3
