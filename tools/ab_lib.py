"""A/B of two BUILDS of the library on one box (compile-time variants): runs a command once per library, alternating, with RSX_LIB
set.  python tools/ab_lib.py <lib a> <lib b> <rounds> -- <command ...>"""
import os
import subprocess
import sys

i = sys.argv.index("--")
a, b, rounds = sys.argv[1], sys.argv[2], int(sys.argv[3])
cmd = sys.argv[i + 1:]
for r in range(rounds):
    for lib in (a, b):
        env = dict(os.environ, RSX_LIB=os.path.abspath(lib))
        out = subprocess.run(cmd, env=env, capture_output=True, text=True)
        last = [l for l in (out.stdout + out.stderr).splitlines() if l.strip() and "amdgpu.ids" not in l]
        print("%-40s | %s" % (os.path.basename(lib), "\n".join(last[-int(os.environ.get("AB_LINES", "1")):])), flush=True)
