"""The host pipeline (bgr_align_all: producer, FASTQ plan, gatherer with its AVX2 line scan, stream workers, fall-back to the exact parser, formatter, ordered
writer) under AddressSanitizer + UBSan and under ThreadSanitizer on RANDOM files with irregular records (tools/irregular_files.py: the files of the
text-route fuzzer) -- tests/test_host_sanitizers.py drives the same harness (tests/sanitize_pipeline.cpp, stand-in devices) over the golden files only.
Runs on the CPU: python tools/fuzz_host_sanitizers.py [seed] [files] [asan|tsan|both].  (Test infrastructure.)"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from synth import Synth
from irregular_files import make_file

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
NF = int(sys.argv[2]) if len(sys.argv) > 2 else 10
which = sys.argv[3] if len(sys.argv) > 3 else "both"
rng = np.random.default_rng(seed)
SRC = os.path.join(ROOT, "bgreat_amd", "csrc")
d = tempfile.mkdtemp(prefix="bgr_fzs_")
exes = {}
for name, flags in (("asan", "-fsanitize=address,undefined"), ("tsan", "-fsanitize=thread")):
    if which not in (name, "both"):
        continue
    exe = os.path.join(d, "sanitize_pipeline_" + name)
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fno-omit-frame-pointer", flags, "-I" + SRC, "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "sanitize_pipeline.cpp"), os.path.join(SRC, "pipeline.cpp"), os.path.join(SRC, "fastx.cpp"), "-o", exe, "-lpthread"])
    exes[name] = exe
env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", TSAN_OPTIONS="halt_on_error=1")
bad = 0
t0 = time.time()
for it in range(NF):
    k = int(rng.choice([5, 15, 21, 31]))
    s = Synth(int(rng.integers(60_000, 300_000)), int(rng.integers(k + 2, 5 * k)), 2, k, 700 + seed + it)
    fastq = bool(rng.random() < 0.45)
    n = int(rng.choice([300, 5_000, 20_003, 45_000]))
    irr = float(rng.choice([0.0, 1e-3, 0.02, 0.2]))
    mixed = bool(rng.random() < 0.5)
    hdr = int(rng.integers(0, 3))
    files = []
    for j in range(int(rng.choice([1, 2]))):
        f = os.path.join(d, "f%d_%d.%s" % (it, j, "fq" if fastq else "fa"))
        make_file(rng, f, s, k, n if j == 0 else max(1, n // 3), fastq, irr, mixed, hdr)
        files.append(f)
    cfg = dict(k=k, fastq=fastq, n=n, irr=irr, mixed=mixed, hdr=hdr, files=len(files))
    for name, exe in exes.items():
        out = os.path.join(d, "out_" + name)
        os.makedirs(out, exist_ok=True)
        p = subprocess.run([exe, "--files", out, str(k), "1" if fastq else "0"] + files, capture_output=True, text=True, env=env, timeout=1500)
        ok = p.returncode == 0 and "files ok" in p.stdout and "ERROR: " not in p.stderr and "WARNING: ThreadSanitizer" not in p.stderr and "runtime error" not in p.stderr
        print("%s %s %s %s" % ("ok      " if ok else "FAILED  ", name, cfg, p.stdout.strip()[-80:]), flush=True)
        if not ok:
            bad += 1
            print(p.stdout[-1500:], p.stderr[-3000:])
            keep = os.path.join(ROOT, "gpurun_out", "fzs_bad_%d_%d" % (seed, it))
            os.makedirs(keep, exist_ok=True)
            for f in files:
                subprocess.call(["cp", f, keep])
    for f in files:
        os.unlink(f)
print("files %d bad %d  %.1fs" % (NF, bad, time.time() - t0))
sys.exit(1 if bad else 0)
