for r in 1 2; do
for l in "" xtnt xtprio xtntprio; do
  if [ -z "$l" ]; then lib=$PWD/ffcnn_amd/lib/libffcnn_hip.so; else lib=$PWD/tools/lab/lib/libffcnn_hip_$l.so; fi
  echo "== ${l:-product}: $(FFCNN_HIP_LIB=$lib python tools/pw_x3t_bench.py one 2>&1 | tail -1 | cut -d: -f2 | cut -d'|' -f1,4)"
done; done
