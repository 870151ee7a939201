"""The host side of libffcnn_hip.so -- ffcnn_host.c: the cfg / .weights parser (restating ffcnn.c:35-247), net_input (259-289),
net_dump, net_free -- under AddressSanitizer + UndefinedBehaviorSanitizer (`make -C ffcnn_amd/csrc san`; SURVEY section 5, VERDICT
r05 item 8).  The device side is a stub (ffcnn_amd/csrc/san/ffgpu_san_stub.c): no GPU, no compute, this is about what the parser
does with text and bytes it did not write.  Corpus: every cfg of the repo, the styles the reference's parser accepts
(tests/test_gpu_cfg_styles.py), truncated .weights files (tests/test_gpu_load_failures.py), and hostile cfgs aimed at what the
reference leaves unchecked -- route / shortcut indices outside the net (ffcnn.c:161-171), oversized value strings (its fixed
`char str[256]`), mask / anchor lists longer than their arrays (ffcnn.c:180-189), sizes whose products overflow an int
(ffcnn.c:149), files without a trailing newline or with NUL bytes.  A sanitizer report aborts the child: the test fails with it."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "ffcnn_amd", "csrc")
BIN = os.path.join(ROOT, "ffcnn_amd", "bin", "ffcnn_host_san")
YCFG = os.path.join(ROOT, "data", "yolo-fastest-1.1.cfg")
YW = os.path.join(ROOT, "data", "yolo-fastest-1.1.weights")


@pytest.fixture(scope="module")
def san():
    subprocess.check_call(["make", "-s", "-C", CSRC, "san"])

    def run(cfg, weights="-", extra=(), env=None):
        e = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1")
        e.update(env or {})
        r = subprocess.run([BIN, cfg, weights] + [str(x) for x in extra], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=120)
        err = r.stderr.decode(errors="replace")
        assert r.returncode == 0 and "Sanitizer" not in err and "runtime error" not in err, "%s %s -> rc %d\n%s" % (cfg, weights, r.returncode, err[-3000:])
        return r.stdout.decode(errors="replace")
    return run


def _cfgs():
    return [YCFG] + [os.path.join(ROOT, "tests", "data", n) for n in sorted(os.listdir(os.path.join(ROOT, "tests", "data"))) if n.endswith(".cfg")]


def test_repo_cfgs_and_net_dump(san):
    want = open(os.path.join(ROOT, "tests", "golden", "net_dump.txt")).read()
    out = san(YCFG, YW)
    assert "net_load: 131 layers, weight_size 356576" in out
    assert want.strip() in out                                 # the reference's own net_dump text (oracle/gen_golden.py)
    for cfg in _cfgs():
        for geo in ((), (0, 0), (64, 64), (33, 95), (640, 448, 1, 1), (320, 320, 2000, 3)):
            assert "net_load:" in san(cfg, "-", geo)
    assert "NULL" in san(YCFG, YW, env={"FFSAN_NODEV": "1"})   # the device side refuses: net_free of a loaded net


def test_cfg_styles(san, tmp_path):
    from test_gpu_cfg_styles import styles
    for src in _cfgs():
        for name, t in styles(open(src).read()):
            p = str(tmp_path / "style.cfg")
            open(p, "w", newline="").write(t)
            assert "layers" in san(p, YW if src == YCFG else "-"), name


@pytest.mark.parametrize("cut", [0, 1, 19, 20, 21, 23, 4096, 4098, 123457, 400001, -3, -4])
def test_truncated_weights(san, tmp_path, cut):
    raw = open(YW, "rb").read()
    p = str(tmp_path / "cut.weights")
    open(p, "wb").write(raw[:cut] if cut >= 0 else raw[:len(raw) + cut])
    assert "131 layers" in san(YCFG, p)
    assert "131 layers" in san(YCFG, str(tmp_path / "missing.weights"))


_NET = "[net]\nwidth=64\nheight=64\nchannels=3\n\n"
_CONV = "[convolutional]\nfilters=8\nsize=3\nstride=1\npad=1\nbatch_normalize=1\nactivation=leaky\n\n"
_HOSTILE = {
    "route far behind": _NET + _CONV + "[route]\nlayers=-7\n\n" + _CONV,
    "route far ahead": _NET + _CONV + "[route]\nlayers=99\n\n" + _CONV,
    "route INT_MIN": _NET + _CONV + "[route]\nlayers=-2147483648, 2147483647\n\n",
    "route five sources": _NET + _CONV + _CONV + "[route]\nlayers=-1,-2,-1,-2,-1,-2,-1\n\n" + _CONV,
    "route to itself": _NET + _CONV + "[route]\nlayers=1\n\n",
    "shortcut outside": _NET + _CONV + "[shortcut]\nfrom=-40\nactivation=linear\n\n" + _CONV,
    "shortcut ahead": _NET + _CONV + "[shortcut]\nfrom=2000000000\nactivation=linear\n\n",
    "value of 5000 characters": _NET + "[convolutional]\nfilters=8\nsize=1\nactivation=" + "leaky" * 1000 + "\n\n[route]\nlayers=" + "-1," * 1700 + "\n\n",
    "yolo lists too long": _NET + _CONV + "[yolo]\nmask=" + ",".join(["7"] * 40) + "\nanchors=" + ",".join(["9"] * 90) + "\nclasses=80\nignore_thresh=.5\n\n",
    "yolo mask out of range": _NET + _CONV + "[yolo]\nmask=-5,9,1000000\nanchors=1,2,3,4\nclasses=2\n\n",
    "huge filters": _NET + "[convolutional]\nfilters=2147483647\nsize=3\nstride=1\npad=1\nactivation=linear\n\n",
    "huge filters x size": _NET + "[convolutional]\nfilters=70000\nsize=181\nstride=1\npad=1\nactivation=linear\n\n",
    "negative filters": _NET + "[convolutional]\nfilters=-8\nsize=3\nstride=1\npad=1\nactivation=linear\n\n" + _CONV,
    "zero size": _NET + "[convolutional]\nfilters=8\nsize=0\nstride=1\npad=1\nactivation=linear\n\n" + _CONV,
    "negative size": _NET + "[convolutional]\nfilters=8\nsize=-3\nstride=1\npad=1\nactivation=linear\n\n" + _CONV,
    "negative stride": _NET + "[convolutional]\nfilters=8\nsize=3\nstride=-1\npad=1\nactivation=linear\n\n[maxpool]\nsize=2\nstride=-2\n\n" + _CONV,
    "groups larger than channels": _NET + "[convolutional]\nfilters=8\nsize=3\ngroups=64\nstride=1\npad=1\nactivation=linear\n\n" + _CONV,
    "negative groups": _NET + "[convolutional]\nfilters=8\nsize=3\ngroups=-2\nstride=1\npad=1\nactivation=linear\n\n" + _CONV,
    "huge input": "[net]\nwidth=2147483647\nheight=2147483647\nchannels=2147483647\n\n" + _CONV,
    "negative input": "[net]\nwidth=-64\nheight=64\nchannels=3\n\n" + _CONV,
    "no net section": _CONV + _CONV,
    "upsample overflow": _NET + _CONV + "[upsample]\nstride=2000000000\n\n" + _CONV,
    "no trailing newline": _NET + "[convolutional]\nfilters=8\nsize=3\nactivation=leaky",
    "header only": "[",
    "unterminated header": _NET + "[convolutional\nfilters=8\n",
    "empty": "",
    "NUL bytes": _NET + "[convolutional]\nfilters=8\x00\nsize=3\n\x00\x00[route]\nlayers=-1\n",
    "only blanks": " \t\r\n \n\n",
    "keys without values": _NET + "[convolutional]\nfilters=\nsize\nstride= \npad==\nactivation=\n\n[yolo]\nmask=\nanchors=,,,\n\n",
    "a thousand layers": _NET + _CONV * 1000,
}


@pytest.mark.parametrize("name", sorted(_HOSTILE))
def test_hostile_cfgs(san, tmp_path, name):
    p = str(tmp_path / "h.cfg")
    open(p, "wb").write(_HOSTILE[name].encode("latin-1"))
    out = san(p, "-")
    out += san(p, YW)                                          # a weights file that does not belong to the cfg
    out += san(p, "-", (96, 32))
    assert "net_load:" in out


def test_seeded_mutations_of_the_real_cfg(san, tmp_path):
    """Byte- and line-level damage to yolo-fastest-1.1.cfg: numbers replaced by extremes, lines dropped / doubled / cut."""
    rng = np.random.default_rng(606)
    lines = open(YCFG).read().splitlines()
    extremes = ["0", "-1", "2147483647", "-2147483648", "99999999999999999999", "1e9", "", "-", "7,7,7,7,7,7,7,7,7,7,7,7", "x"]
    for it in range(40):
        ls = list(lines)
        for _ in range(int(rng.integers(1, 6))):
            i = int(rng.integers(0, len(ls)))
            op = int(rng.integers(0, 5))
            if op == 0 and "=" in ls[i]:
                ls[i] = ls[i].split("=")[0] + "=" + extremes[int(rng.integers(0, len(extremes)))]
            elif op == 1:
                del ls[i]
            elif op == 2:
                ls.insert(i, ls[i])
            elif op == 3:
                ls[i] = ls[i][:int(rng.integers(0, len(ls[i]) + 1))]
            else:
                ls[i] = re.sub(r"-?\d+", lambda m: extremes[int(rng.integers(0, 5))], ls[i])
        p = str(tmp_path / "m.cfg")
        open(p, "w").write("\n".join(ls) + "\n")
        assert "net_load:" in san(p, YW), "mutation %d" % it
