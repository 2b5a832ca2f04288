"""Where does score_topk_kernel spend its time?  Builds copies of the library in which one
part of the kernel is cut out (results are then wrong on purpose) and times each copy.

  python tools/eval_variants.py build      # here, on the CPU host (hipcc cross-compiles)
  python tools/eval_variants.py run        # on the GPU box
"""
import os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from recommend_amd import build as B

OUT = os.path.join(ROOT, "tools", "_variants")

# s_memtime stamps and event counts of one wavefront, printed by a few wavefronts
STAMP = [
    ("  LaneList<KPAD> list;\n", "  uint64_t st_sel = 0, st_flush = 0, st_score = 0; int n_slow = 0, n_flush = 0, n_rounds = 0;\n  const uint64_t st0 = __builtin_readcyclecounter();\n  LaneList<KPAD> list;\n"),
    ("      list.insert(v, vid);\n", "      list.insert(v, vid);\n      ++n_rounds;\n"),
    ("    if (!__any(m > thr)) return false;\n", "    if (!__any(m > thr)) return false;\n    ++n_slow;\n"),
    ("      if (full) flush();\n", "      if (full) { const uint64_t f0 = __builtin_readcyclecounter(); flush(); st_flush += __builtin_readcyclecounter() - f0; ++n_flush; }\n"),
    ("    int g_next = 0;\n", "    const uint64_t c3 = __builtin_readcyclecounter();\n    int g_next = 0;\n"),
    ("    } while (g_next < 8);\n", "    } while (g_next < 8);\n    st_sel += __builtin_readcyclecounter() - c3;\n"),
    ("    if ((t - t_lo) & 1)\n", "    const uint64_t c0 = __builtin_readcyclecounter();\n    if ((t - t_lo) & 1)\n"),
    ("    const int32_t item0 = t * TI;\n    // rows past", "    st_score += __builtin_readcyclecounter() - c0;\n    const int32_t item0 = t * TI;\n    // rows past"),
    ("  flush();\n\n  if (user_ok) {", "  flush();\n  if (lane == 0 && wave == 1 && blockIdx.x % 701 == 3)\n"
     "    printf(\"blk %d tiles %d total %llu score %llu sel %llu flush %llu n_slow %d n_flush %d n_rounds %d\\n\", (int)blockIdx.x, (int)(t_hi - t_lo),\n"
     "           (unsigned long long)(__builtin_readcyclecounter() - st0), (unsigned long long)st_score, (unsigned long long)st_sel,\n"
     "           (unsigned long long)st_flush, n_slow, n_flush, n_rounds);\n\n  if (user_ok) {"),
]
VARIANTS = {
    "base": [],
    "stamp": STAMP,
}


def build():
    os.makedirs(OUT, exist_ok=True)
    B.build()
    src = open(os.path.join(B.CSRC, "eval.hip")).read()
    procs = []
    for name, subs in VARIANTS.items():
        s = src
        for old, new in subs:
            assert old in s, (name, old[:40])
            s = s.replace(old, new)
        path = os.path.join(OUT, f"eval_{name}.hip")
        open(path, "w").write(s)
        obj = path + ".o"
        cmd = ["/opt/rocm/bin/hipcc", "-x", "hip"] + [f for f in B.FLAGS if f != "-shared"] + \
              ["-I", B.CSRC, "-c", path, "-o", obj]
        procs.append((name, obj, subprocess.Popen(cmd, stderr=subprocess.DEVNULL)))
    for name, obj, p in procs:
        assert p.wait() == 0, name
        others = [os.path.join(B.LIB_DIR, f + ".o") for f in B.SOURCES if f != "eval.hip"]
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(OUT, f"librk_{name}.so"), obj] + others)
        os.remove(obj)
    print("built", list(VARIANTS))


def run():
    names = [n for n in VARIANTS if not os.environ.get("ONLY") or n in os.environ["ONLY"].split(",")]
    for name in names:
        env = dict(os.environ, RECOMMEND_AMD_LIB=os.path.join(OUT, f"librk_{name}.so"))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "eval_probe.py")] + sys.argv[2:],
                           env=env, capture_output=True, text=True, timeout=300)
        for l in r.stdout.splitlines():
            if l.startswith("blk"):
                print("   ", l)
        line = [l for l in r.stdout.splitlines() if l.startswith("score_topk")]
        print(f"{name:16s} {line[-1] if line else r.stderr[-300:]}", flush=True)


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
