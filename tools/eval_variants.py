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

OCC1 = ("__launch_bounds__(256, DC == 1 ? 2 : 1)", "__launch_bounds__(256, 1)")
FAST = "    if (!__any(m > thr)) return;\n"
VARIANTS = {
    "base": [],
    "occ1": [OCC1],
    "nofast": [(FAST, "")],
    "ns4": [("  return best;\n}", "  return 4;\n}")],
    "ns1": [("  return best;\n}", "  return 1;\n}")],
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
        line = [l for l in r.stdout.splitlines() if l.startswith("score_topk")]
        print(f"{name:16s} {line[-1] if line else r.stderr[-300:]}", flush=True)


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
