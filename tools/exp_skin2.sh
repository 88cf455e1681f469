#!/bin/bash
source <(sed -n '/^run()/,/^}/p' tools/exp_skin.sh)
run hz2_049 hz2 0.49 ""
run hz2_058 hz2 0.578 ""
run hz2_067 hz2 0.668 ""
run main_041 main 0.41 ""
run main_058 main 0.578 ""
