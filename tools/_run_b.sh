for i in 1 2; do
python tools/pw_x3t_bench.py one 2>&1 | grep -v amdgpu | cut -c1-200
FFCNN_HIP_LIB=$PWD/tools/lab/lib/libffcnn_hip_splitfirst.so python tools/pw_x3t_bench.py one 2>&1 | grep -v amdgpu | cut -c1-200
done
