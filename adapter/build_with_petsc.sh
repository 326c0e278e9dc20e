#!/usr/bin/env bash
# adapter/build_with_petsc.sh -- the first build of adapter/petiga_amd_petsc.c against a real PETSc + PetIGA tree, and the
# end-to-end check that closes SURVEY 8f-2: demo/Poisson3D solved once with PetIGA's own CPU assembly and once with the engine
# behind the unchanged IGAComputeSystem, the two KSP residual histories compared.
#
# This image has no PETSc, so the script has never run here; it is the recipe for the first box that has
#   PETSC_DIR, PETSC_ARCH   a PETSc >= 3.19 configured --with-hip (MATAIJHIPSPARSE; COO assembly on the device)
#   PETIGA_DIR              a checkout of dalcinl/PetIGA (its own build system is used untouched)
#   IGX_DIR                 this repository (default: the parent of this script), with petiga_amd/libpetiga_amd.so built
# Steps (each stops the script on failure):
#   0. syntax-only compile of the adapter against the real headers  (signature drift shows up here first)
#   1. PetIGA built twice from the same checkout: as it is (CPU reference) and with the adapter
#      (-DPETIGA_HAVE_AMD: the seven IGACompute* bodies renamed to IGACompute*_CPU, the adapter providing the public names)
#   2. demo/Poisson3D -iga_elements N -iga_degree 3 -ksp_monitor, both builds; the device build with -iga_mat_type aijhipsparse
#      -iga_vec_type hip; residual histories compared to RTOL (default 1e-8 relative per iteration)
#   3. the same with mpiexec -n 2 when MPIEXEC is set (ghost rows through PETSc's own COO assembly)
set -euo pipefail
: "${PETSC_DIR:?set PETSC_DIR}"; : "${PETSC_ARCH:?set PETSC_ARCH}"; : "${PETIGA_DIR:?set PETIGA_DIR (a PetIGA checkout)}"
IGX_DIR="${IGX_DIR:-$(cd "$(dirname "$0")/.." && pwd)}"
N="${N:-32}"; RTOL="${RTOL:-1e-8}"; WORK="${WORK:-$(mktemp -d /tmp/igx_petsc.XXXXXX)}"
test -f "$IGX_DIR/petiga_amd/libpetiga_amd.so" || { echo "build the library first: python -c 'import __graft_entry__ as g; g.build()'"; exit 2; }
PETSC_CC_INCLUDES="-I$PETSC_DIR/include -I$PETSC_DIR/$PETSC_ARCH/include"
CC="${CC:-$(grep -E '^CC *=' "$PETSC_DIR/$PETSC_ARCH/lib/petsc/conf/petscvariables" | head -1 | cut -d= -f2-)}"
CC="${CC:-mpicc}"

echo "== 0. syntax-only compile of the adapter against PETSc's and PetIGA's headers"
$CC -fsyntax-only -Wall -Wextra -DPETIGA_HAVE_AMD $PETSC_CC_INCLUDES -I"$PETIGA_DIR/include" -I"$IGX_DIR/include" "$IGX_DIR/adapter/petiga_amd_petsc.c"

echo "== 1a. PetIGA as it is (CPU reference) -> $WORK/cpu"
rm -rf "$WORK/cpu" "$WORK/amd"; mkdir -p "$WORK"
cp -r "$PETIGA_DIR" "$WORK/cpu"
make -C "$WORK/cpu" PETSC_DIR="$PETSC_DIR" PETSC_ARCH="$PETSC_ARCH" PETIGA_DIR="$WORK/cpu" all

echo "== 1b. PetIGA with the adapter -> $WORK/amd"
cp -r "$PETIGA_DIR" "$WORK/amd"
# the seven driver bodies keep living as IGACompute*_CPU (the adapter's fall-through); the public names come from the adapter
for f in petigaksp.c petigasnes.c petigats.c; do
  sed -E -i 's/^PetscErrorCode (IGACompute(Vector|Matrix|System|Function|Jacobian|IFunction|IJacobian))\(/PetscErrorCode \1_CPU(/' "$WORK/amd/src/$f"
done
cp "$IGX_DIR/adapter/petiga_amd_petsc.c" "$WORK/amd/src/petigaamd.c"
cp "$IGX_DIR/include/petiga_amd.h" "$WORK/amd/include/"
# PetIGA's makefiles take the source list from src/makefile (SOURCEC) -- add the new unit there
sed -E -i 's/^(SOURCEC *=)/\1 petigaamd.c/' "$WORK/amd/src/makefile"
make -C "$WORK/amd" PETSC_DIR="$PETSC_DIR" PETSC_ARCH="$PETSC_ARCH" PETIGA_DIR="$WORK/amd" \
     CFLAGS="-DPETIGA_HAVE_AMD" CLINKER_SLFLAG="-Wl,-rpath," OTHERSHAREDLIBS="-L$IGX_DIR/petiga_amd -lpetiga_amd -Wl,-rpath,$IGX_DIR/petiga_amd" all

echo "== 2. demo/Poisson3D: CPU assembly vs the engine behind IGAComputeSystem"
# the demo registers its host callback with IGASetFormSystem; the device build also names the built-in form that stands for it
# (one line in main(): IGASetFormAMD(iga,IGX_FORM_POISSON,(PetscReal[]){1.0},1) under #if defined(PETIGA_HAVE_AMD))
grep -q IGASetFormAMD "$WORK/amd/demo/Poisson3D.c" || sed -E -i 's/^( *)(ierr = IGASetFormSystem\(iga,System,NULL\);CHKERRQ\(ierr\);)/\1\2\n#if defined(PETIGA_HAVE_AMD)\n\1{ PetscReal one = 1.0; ierr = IGASetFormAMD(iga,IGX_FORM_POISSON,\&one,1);CHKERRQ(ierr); }\n#endif/' "$WORK/amd/demo/Poisson3D.c"
for w in cpu amd; do make -C "$WORK/$w/demo" PETSC_DIR="$PETSC_DIR" PETSC_ARCH="$PETSC_ARCH" PETIGA_DIR="$WORK/$w" Poisson3D; done
COMMON="-iga_elements $N -iga_degree 3 -ksp_type cg -pc_type jacobi -ksp_rtol 1e-10 -ksp_monitor"
run() { ( cd "$WORK/$1/demo" && ${2:-} ./Poisson3D $COMMON ${3:-} ) | grep 'KSP Residual norm' | awk '{print $NF}'; }
run cpu "" "" > "$WORK/hist_cpu.txt"
run amd "" "-iga_mat_type aijhipsparse -iga_vec_type hip" > "$WORK/hist_amd.txt"
python3 - "$WORK/hist_cpu.txt" "$WORK/hist_amd.txt" "$RTOL" <<'PY'
import sys
a = [float(x) for x in open(sys.argv[1])]; b = [float(x) for x in open(sys.argv[2])]; tol = float(sys.argv[3])
assert a and len(a) == len(b), "different iteration counts: %d vs %d" % (len(a), len(b))
worst = max(abs(x - y) / max(abs(x), 1e-300) for x, y in zip(a, b))
print("KSP residual histories: %d iterations, worst relative difference %.3e (tolerance %g)" % (len(a), worst, tol))
sys.exit(0 if worst < tol else 1)
PY

if [ -n "${MPIEXEC:-}" ]; then
  echo "== 3. two ranks (ghost rows through PETSc's COO assembly)"
  run cpu "$MPIEXEC -n 2" "" > "$WORK/hist_cpu2.txt"
  run amd "$MPIEXEC -n 2" "-iga_mat_type aijhipsparse -iga_vec_type hip" > "$WORK/hist_amd2.txt"
  python3 - "$WORK/hist_cpu2.txt" "$WORK/hist_amd2.txt" "$RTOL" <<'PY'
import sys
a = [float(x) for x in open(sys.argv[1])]; b = [float(x) for x in open(sys.argv[2])]; tol = float(sys.argv[3])
assert a and len(a) == len(b)
worst = max(abs(x - y) / max(abs(x), 1e-300) for x, y in zip(a, b))
print("2 ranks: worst relative difference %.3e" % worst); sys.exit(0 if worst < tol else 1)
PY
fi
echo "adapter verified against $PETIGA_DIR: $WORK"
