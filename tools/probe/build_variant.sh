# measurement builds with extra -D flags for SEVERAL source files: build/libsad_<name>.so
# usage: bash tools/probe/build_variant.sh noscan "-DSAD_NOSCAN" mlp_reg.hip mlp_coop.hip
name=$1; defs=$2; shift 2
cd "$(dirname "$0")/../../3dsad-main_amd/csrc" || exit 1
mkdir -p ../../build
F="-O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -fvisibility=hidden -std=c++17 -Wno-unused-function"
objs=""
for f in *.hip; do
  case " $* " in
    *" $f "*)
      X=""; case $f in mlp_reg.hip|mlp_coop.hip|mlp_bf16_reg.hip) X="-mllvm -amdgpu-mfma-vgpr-form -mllvm -pragma-unroll-threshold=4000000";; esac
      /opt/rocm/bin/hipcc $F $X $defs -c $f -o ../../build/${name}_${f%.hip}.o || exit 1
      objs="$objs ../../build/${name}_${f%.hip}.o";;
    *) objs="$objs ${f%.hip}.o";;
  esac
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/libsad_${name}.so $objs
