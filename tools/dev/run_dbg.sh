mkdir -p gpurun_out/r03
for t in "tests/test_gpu_mesh_adversarial.py -k n_sv2" "tests/test_gpu_mesh_adversarial.py -k n_sv3" "tests/test_gpu_mesh_adversarial.py -k n_sv0" "tests/test_gpu_mesh_adversarial.py -k n_sv1" "tests/test_gpu_production.py" "tests/test_gpu_parity.py -k mesh"; do
  echo "== $t"; timeout -k 5 200 python -m pytest $t -x -q -m gpu 2>&1 | tail -4
done > gpurun_out/r03/dbg_a3.txt 2>&1
cat gpurun_out/r03/dbg_a3.txt
