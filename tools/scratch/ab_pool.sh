for v in 0 1; do
CHEM_TRACE=1 python bench.py --f64-steps 0 --cpu-steps 0 --steps 2000 --late-stage 0 --no-roofline > gpurun_out/r3_tr$v.json 2> gpurun_out/r3_tr$v.err
echo "run $v: $(grep -o 'reaction_step_ms[^,}]*' gpurun_out/r3_tr$v.json) $(grep 'host events' gpurun_out/r3_tr$v.err | tr '\n' ' ')"
done
