#!/bin/bash
# Soak of the in-process group (one host thread per member in the step loop): the group tests N times in a row, the failures counted.  Why: round 5 saw one
# abort in eight runs with a host thread per member INSIDE MLTInit (profiles/r05_final_note_group_init_abort.txt); the init has run on the calling thread
# since, and the one allocation the threaded step loop made (the relocation's staging cut) is serialised since round 6.   usage: scripts/debug/group_soak.sh [runs] > out.txt  (GPU)
N=${1:-12}; fail=0
for i in $(seq $N); do
  timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "group_of_ranks" > /tmp/soak_$i.log 2>&1 || { fail=$((fail+1)); tail -5 /tmp/soak_$i.log; }
  tail -1 /tmp/soak_$i.log
done
echo "group soak: $N runs, $fail failed"
