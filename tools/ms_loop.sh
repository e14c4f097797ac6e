tot=0
for i in 1 2 3 4; do
  n=$(python tools/ms_stage_check2.py 4 3 4 3 2 1 4 3 2>&1 | grep -c "differing: \[(")
  tot=$((tot+n))
done
echo "stage runs mismatching: $tot of 128"
for i in 1 2 3; do python tools/ms_check.py 2>&1 | grep -c "mismatches: \[("; done
