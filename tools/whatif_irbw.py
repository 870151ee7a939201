#!/usr/bin/env python3
"""What-if builds of the fused-block kernels (WRONG results, timing only): out-of-tree copies of ffcnn_amd/csrc with one piece of the
per-group work of k_irbw / k_irbw2 removed, placed at ffcnn_amd/lib/libffcnn_hip_wi_<name>.so; time them with
  FFCNN_HIP_LIB=$PWD/ffcnn_amd/lib/libffcnn_hip_wi_<name>.so python bench.py --no-extras --no-node-line --no-cpu-baseline --no-kernel-roofline
variants: mask (no border select), exp1 (one expand k-step instead of KS1), dw13 (one of three tap columns), proj1 (half of the project
MFMAs), noldsw (E-slice stores predicated off).  usage: python tools/whatif_irbw.py <variant>   (run here; no GPU needed)"""
import re,sys,os,shutil,subprocess
R='/root/repo'
variants={
 'mask': [(r'const bool in = \(mb >> \(si \* 4 \+ q\)\) & 1u;', 'const bool in = true;')],
 'exp1': [(r'for \(int ks = 0; ks < KS1; ks\+\+\) \{(\s+a4\[)', r'for (int ks = 0; ks < 1; ks++) {\1')],
 'dw13': [(r'for \(int kx = 0; kx < 3; kx\+\+\) dv', 'for (int kx = 0; kx < 1; kx++) dv')],
 'proj1': [(r'for \(int j = 0; j < 2; j\+\+\) \{(\s+const float a = w2)', r'for (int j = 0; j < 1; j++) {\1'), (r'for \(int j = 0; j < 2; j\+\+\)(\s+#pragma unroll\s+for \(int t = 0; t < OT; t\+\+\))', r'for (int j = 0; j < 1; j++)\1')],
 'noldsw': [(r'\*reinterpret_cast<v4f_w \*>\(dst\) = \(v4f_w\)\{ y\[0\]\.x, y\[0\]\.y, y\[1\]\.x, y\[1\]\.y \};', 'if (y[0].x == 1234.5f) *reinterpret_cast<v4f_w *>(dst) = (v4f_w){ y[0].x, y[0].y, y[1].x, y[1].y };'),
            (r'\*reinterpret_cast<v4f_w \*>\(dst \+ NQ \* 4\) = \(v4f_w\)\{ y\[2\]\.x, y\[2\]\.y, y\[3\]\.x, y\[3\]\.y \};', 'if (y[2].x == 1234.5f) *reinterpret_cast<v4f_w *>(dst + NQ * 4) = (v4f_w){ y[2].x, y[2].y, y[3].x, y[3].y };')],
}
name=sys.argv[1]
d='/tmp/wi/'+name
shutil.rmtree(d,ignore_errors=True)
os.makedirs(d+'/ffcnn_amd')
shutil.copytree(R+'/ffcnn_amd/csrc', d+'/ffcnn_amd/csrc', ignore=shutil.ignore_patterns('build'))
shutil.copytree(R+'/include', d+'/include')
n=0
for f in ('ffgpu_irb_wave.inc','ffgpu_irb_wave2.inc'):
    p=d+'/ffcnn_amd/csrc/'+f
    s=open(p).read()
    for a,b in variants[name]:
        s,k=re.subn(a,b,s); n+=k
    open(p,'w').write(s)
print(name,'substitutions',n)
r=subprocess.run(['make','-C',d+'/ffcnn_amd/csrc','../lib/libffcnn_hip.so'],capture_output=True,text=True)
if r.returncode: print(r.stderr[-800:])
else: shutil.copy(d+'/ffcnn_amd/lib/libffcnn_hip.so', R+'/ffcnn_amd/lib/libffcnn_hip_wi_%s.so'%name); print('built',name)
