set -e
cd "$(dirname "$0")/../multigridcmt_amd/csrc"
rm -rf ../../build/variants ../../variants; mkdir -p ../../build/variants ../../variants
build() { name=$1; shift; make -s -j8 OUT=$PWD/../../variants/lib_$name.so OBJDIR=$PWD/../../build/variants/obj_$name "$@"; echo built $name; }
build wide_d3 EXTRA=-DMGCMT_FUSED_WIDE_DEPTH=3
