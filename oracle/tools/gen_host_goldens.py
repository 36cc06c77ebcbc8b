#!/usr/bin/env python3
"""Generate tests/golden/host_tables.json with the reference's OWN loader + camera (container-only).

TEST INFRASTRUCTURE.  Runs oracle/_ref/dump_tables (the unmodified reference sources compiled by
oracle/ref/Makefile) on every scenefile of the reference checkout and records the uniform tables it
produces.  The scenefiles themselves (JSON data, the reference's manual test fixtures) are
copied next to the expected tables as inputs: tests/golden/scenes/<dir>/<file>.json.
Run here only:  python oracle/tools/gen_host_goldens.py
"""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/scenefiles"
BIN = os.path.join(ROOT, "oracle", "_ref", "dump_tables")
OUT = os.path.join(ROOT, "tests", "golden")
SIZES = [(1024, 768), (3840, 2160)]


def main():
    if not os.path.exists(BIN):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle", "ref")])
    env = dict(os.environ, LD_LIBRARY_PATH="/usr/lib/x86_64-linux-gnu:/opt/conda/lib")
    golden = {}
    for sub in sorted(os.listdir(REF)):
        d = os.path.join(REF, sub)
        if not os.path.isdir(d):
            continue
        for fn in sorted(os.listdir(d)):
            if not fn.endswith(".json"):
                continue
            rel = f"{sub}/{fn}"
            os.makedirs(os.path.join(OUT, "scenes", sub), exist_ok=True)
            shutil.copyfile(os.path.join(d, fn), os.path.join(OUT, "scenes", sub, fn))
            entry = {}
            for (w, h) in SIZES:
                p = subprocess.run([BIN, os.path.join(d, fn), str(w), str(h)], env=env, capture_output=True, text=True)
                line = [l for l in p.stdout.splitlines() if l.startswith("@@JSON ")]
                if p.returncode != 0 or not line:
                    entry = {"ok": False, "crashed": True}
                    break
                t = json.loads(line[-1][7:])
                if not t["ok"]:
                    entry = {"ok": False}
                    break
                if not entry:
                    entry = {k: v for k, v in t.items() if k not in ("view", "proj", "invProjView")}
                    for o in entry["objects"]:  # resolved against the scenefile's grand-parent directory: keep the tail
                        o["textureFile"] = os.path.relpath(o["textureFile"], os.path.dirname(d)) if o["textureFile"] else ""
                    entry["camera"] = {}
                entry["camera"][f"{w}x{h}"] = {k: t[k] for k in ("view", "proj", "invProjView")}
            golden[rel] = entry
            print(rel, "ok" if entry.get("ok") else "REJECTED", file=sys.stderr)
    with open(os.path.join(OUT, "host_tables.json"), "w") as f:
        json.dump(golden, f, indent=0, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
