"""GPU box: wall / user / sys CPU seconds of the reference and the hooked encoder on the same clip (are waiting threads spinning?)"""
import os
import subprocess
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import svtlib as S
S.write_clip("/tmp/c.yuv", "motion", 3840, 2160, 16, 7)
args = "-i /tmp/c.yuv -w 3840 -h 2160 -n 96 -nb 16 -encMode 7 -pred-struct 2 -hierarchical-levels 2 -sao 1 -fps 60 -q 32 -asm 1 -b /tmp/o.265".split()
for app in ("oracle/_ref/SvtHevcEncApp_ref", "integration/_build/SvtHevcEncApp_hip"):
    for extra in ({}, {"SVT_HOOK_ENCODEPASS": "1"}) if "hip" in app else ({},):
        t0 = time.time()
        p = subprocess.Popen([os.path.join(ROOT, app)] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=dict(os.environ, **extra))
        out = p.stdout.read().decode()
        _, st, ru = os.wait4(p.pid, 0)
        fps = [l.strip() for l in out.splitlines() if "Average" in l or "Total" in l or "Max" in l]
        print(app, extra, "wall %.2f user %.1f sys %.1f vol-cs %d invol-cs %d | %s" % (time.time() - t0, ru.ru_utime, ru.ru_stime, ru.ru_nvcsw, ru.ru_nivcsw, fps))
