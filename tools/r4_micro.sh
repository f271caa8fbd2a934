#!/bin/bash
set -u
bash tools/ab_lib.sh r04f_c4 3 "--workload c4 --steps 20 --warmup 5" - wflat brcp || exit 1
