# measurement builds: build/libsad_<name>.so with extra -D flags for ONE source file.
# usage: bash tools/probe/build_stamps.sh cstamps mlp_coop.hip -DSAD_COOP_STAMPS
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../../3dsad-main_amd/csrc" || exit 1
mkdir -p ../../build
F="-O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -fvisibility=hidden -std=c++17 -Wno-unused-function"
X=""
case $src in mlp_reg.hip|mlp_coop.hip|mlp_bf16_reg.hip) X="-mllvm -amdgpu-mfma-vgpr-form -mllvm -pragma-unroll-threshold=4000000";; esac
/opt/rocm/bin/hipcc $F $X "$@" -c $src -o ../../build/${name}_${src%.hip}.o || exit 1
objs=""
for f in *.hip; do [ "$f" = "$src" ] || objs="$objs ${f%.hip}.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/libsad_${name}.so $objs ../../build/${name}_${src%.hip}.o
