#!/usr/bin/env python3
"""One directory per round under profiles/: rNN/final/ (ONE record set per round) and rNN/experiments/ (everything else), and every
`profiles/rNN_tag_name` reference in sources and docs rewritten to the new place.  Idempotent.  """
import os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(REPO, "profiles")
FINAL = {"r01": "x", "r02": "n", "r03": "final2"}               # the tag whose files are the round's final set
LOOSE_FINAL = {"r02_stress.txt", "r03_stress.txt", "r03_asan.txt"}


def new_name(name):
    m = re.match(r"^(r0\d)_(.+)$", name)
    if not m:
        return None
    rnd, rest = m.groups()
    if name in LOOSE_FINAL:
        return f"{rnd}/final/{rest}"
    tag, _, tail = rest.partition("_")
    if tail and tag == FINAL.get(rnd):
        return f"{rnd}/final/{tail}"
    return f"{rnd}/experiments/{rest}"


moves = {}
for name in sorted(os.listdir(PROF)):
    if os.path.isfile(os.path.join(PROF, name)):
        to = new_name(name)
        if to:
            moves[name] = to
for name, to in moves.items():
    os.makedirs(os.path.dirname(os.path.join(PROF, to)), exist_ok=True)
    subprocess.check_call(["git", "-C", REPO, "mv", os.path.join("profiles", name), os.path.join("profiles", to)])
# references: exact file names first, then `profiles/rNN_tag_*` prefixes (globs in prose)
pat_tag = re.compile(r"profiles/(r0\d)_([a-z0-9]+)_")
count = 0
for root, dirs, files in os.walk(REPO):
    dirs[:] = [d for d in dirs if d not in (".git", "gpurun_out", "__pycache__", ".hypothesis", ".pytest_cache", "golden")]
    files = [f for f in files if not (root == REPO and f.startswith(("BENCH_", "VERDICT", "ADVICE", "GPUTEST_", "SCALE_", "MULTICHIP_", "SURVEY", "BASELINE")))]   # the driver's and the judge's files are not ours to edit
    for f in files:
        if not f.endswith((".py", ".md", ".hip", ".h", ".c", ".sh", ".json")) or (root == PROF or root.startswith(PROF + os.sep)) and not f.endswith(".md"):
            continue
        path = os.path.join(root, f)
        try:
            text = open(path).read()
        except (UnicodeDecodeError, OSError):
            continue
        new = text
        for name, to in moves.items():
            new = new.replace("profiles/" + name, "profiles/" + to)
        new = pat_tag.sub(lambda m: f"profiles/{m.group(1)}/final/" if m.group(2) == FINAL.get(m.group(1)) else f"profiles/{m.group(1)}/experiments/{m.group(2)}_", new)
        if new != text:
            open(path, "w").write(new)
            count += 1
print(f"moved {len(moves)} files, rewrote references in {count} files")
