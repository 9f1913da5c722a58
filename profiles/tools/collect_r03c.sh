#!/bin/bash
# session 3 of round 3: rocprofv3 stats + counters of the final build for the two larger single-GPU workloads
bash profiles/tools/collect.sh r03c_c2 512x512x256/smag/nsv1 67108864 --size 512x512x256 --sgs smag --nsv 1
bash profiles/tools/collect.sh r03c_c3 1024x512x512/vreman/nsv0 268435456 --size 1024x512x512 --steps 4
