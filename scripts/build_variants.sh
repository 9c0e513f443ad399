set -e
cd "$(dirname "$0")/../multigridcmt_amd/csrc"
rm -rf ../../build/variants; mkdir -p ../../build/variants
build() { name=$1; shift; make -s -j8 OUT=$PWD/../../build/variants/lib_$name.so OBJDIR=$PWD/../../build/variants/obj_$name "$@"; echo built $name; }
build fma FUSED_FLAGS=-ffp-contract=fast
