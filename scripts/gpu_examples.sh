# the counterparts of the reference's driver scripts, end to end on the GPU
set -x
mkdir -p gpurun_out
for e in unit_tests pot_matrix_vcycle_1d pot_matrix_vcycle_2d pot_matrix_vcycle_2d_resident potential_well_rq_2d rqmin; do
  timeout -k 10 200 python examples/$e.py > gpurun_out/example_$e.log 2>&1; echo "$e rc=$?"; tail -4 gpurun_out/example_$e.log
done
