export FFCNN_HIP_LIB=$PWD/tools/lab/lib/libffcnn_hip_trace.so
IRB_TRACE_CONCURRENT=1 python tools/irb_trace.py 2>&1 | grep "trace\|^irbw" | cut -c1-400 > gpurun_out/irb_trace_conc.txt
cat gpurun_out/irb_trace_conc.txt | tail -40
