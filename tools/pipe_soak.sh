for i in 1 2 3 4 5 6 7 8; do
  timeout -k 10 120 python tools/pipe_fault_repro.py 2 > gpurun_out/soak_$i.txt 2>&1
  rc=$?
  f=$(grep -c "Memory access fault" gpurun_out/soak_$i.txt)
  echo "run $i rc=$rc faults=$f"
  if [ "$f" != "0" ] || [ "$rc" != "0" ]; then exit 1; fi
done
