set -o pipefail
R=$PWD; O=$R/gpurun_out; export TMPDIR=/tmp
run() { # label, env, coarse, cycles
  env $2 python3 bench.py --workload pmg_annulus --nref 8 --coarse $3 --coarse-cycles $4 --no-cpu-baseline --no-secondary > $O/r3h_$1.json 2> $O/r3h_$1.err || { tail -5 $O/r3h_$1.err; return 1; }
  python3 -c "
import json
d=json.load(open('$O/r3h_$1.json')); print('$1', round(d['ms_per_step'],3), 'its', d['config']['cg_iterations_reltol_1e-4'], 'cg_throughput', '%.3e' % d['config']['cg_throughput_dofs_x_iterations_per_s'])"
}
run amg_d2_c1 A=1 amg 1 && run amg_d2_c2 A=1 amg 2 && run amg_d3_c1 MGAMD_AMG_SMOOTHER_DEGREE=3 amg 1 && run amg_d3_c2 MGAMD_AMG_SMOOTHER_DEGREE=3 amg 2 && run gmg_c1 A=1 gmg_vcycle 1 && run gmg_c2 A=1 gmg_vcycle 2 && run cgamg_d3 MGAMD_AMG_SMOOTHER_DEGREE=3 cg_with_amg 1
