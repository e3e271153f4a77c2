# Host code under AddressSanitizer / UBSan, CPU only (GPU ASan is not available on this pool):
#   1. the CPU oracle (gcc -fsanitize=address,undefined) through its pin, math and adversarial tests,
#   2. the host side of libdsim_hip.so (index builders, clearance map, fine grid, writers, RNG seeding, argument checks;
#      hipcc -fsanitize=address applies to the host half only) through the whole `-m "not gpu"` suite.
# usage: bash tools/asan_cpu.sh     (from the repo root; builds into /tmp, leaves the tree's libraries alone)
set -e
T=$(mktemp -d)
F="-O1 -g -fPIC -shared -ffp-contract=off -mfma -fno-math-errno -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer"
mkdir -p $T/oracle
gcc $F oracle/dsim_oracle.c -o $T/oracle/libdsim_oracle.so -lm
gcc $F -DDSO_SHARED_MATH -Idisimpy_amd/csrc oracle/dsim_oracle.c -o $T/oracle/libdsim_oracle_exact.so -lm
echo "== oracle under ASan + UBSan"
ASAN_OPTIONS=detect_leaks=0 DSIM_ORACLE_DIR=$T/oracle LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
  python -m pytest tests/test_oracle_pins.py tests/test_mesh_adversarial.py tests/test_dsim_math.py -q
RT=$(find /opt/rocm/lib/llvm -name "libclang_rt.asan-x86_64.so" | head -1)
(cd disimpy_amd/csrc && /opt/rocm/bin/hipcc -O1 -g -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -fsanitize=address \
  -fno-omit-frame-pointer -shared-libsan -Wno-option-ignored dsim_hip.hip -o $T/libdsim_hip_asan.so)
echo "== host half of libdsim_hip.so under ASan"
ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 LD_PRELOAD=$RT DSIM_LIB=$T/libdsim_hip_asan.so python -m pytest tests -q -m "not gpu" -k "not gloo and not codeobj"
rm -rf $T
