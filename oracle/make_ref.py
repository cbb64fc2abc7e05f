"""oracle/_ref: the reference's OWN Python modules for this path, placed where the GPU box can import them.  TEST INFRASTRUCTURE.

    python oracle/make_ref.py          (build container only: needs /root/reference; also run by __graft_entry__.build())

/root/reference does not exist on the GPU box and reference sources are never committed to this repository.  The path has no C / C++
to compile into a library, so the "reference build" of this tier is a scratch copy of the Python modules the
path runs - guided_diffusion/*.py (UNetModel, GaussianDiffusion, respace), score_sde/sde_lib.py and score_sde/models/*.py (NCSNpp) -
copied from where they lie under /root/reference into oracle/_ref/ (listed in .gitignore: stays out of history; NOT in .gpurunignore:
travels with the snapshot like the built .so files).  Only the sha256 of every file is tracked (oracle/ref_modules.sha256);
oracle/ref_loader.py refuses to import a copy whose digests differ.  Users: bench.py's `cpu_baseline` leg (`kind: "reference"` when the
copy is present and verified, else the oracle restatement, `kind: "port"`), tests/test_oracle_golden.py."""
import glob
import hashlib
import os
import shutil

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference"
DST = os.path.join(HERE, "_ref")
SHA = os.path.join(HERE, "ref_modules.sha256")
PATTERNS = ("guided_diffusion/*.py", "score_sde/sde_lib.py", "score_sde/models/*.py")


def main():
    if not os.path.isdir(SRC):
        print("reference checkout not present: oracle/_ref not (re)generated")
        return False
    lines = []
    for pat in PATTERNS:
        for src in sorted(glob.glob(os.path.join(SRC, pat))):
            rel = os.path.relpath(src, SRC)
            dst = os.path.join(DST, rel)
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            shutil.copyfile(src, dst)
            lines.append(f"{hashlib.sha256(open(dst, 'rb').read()).hexdigest()}  {rel}\n")
    text = "".join(lines)
    if not os.path.exists(SHA) or open(SHA).read() != text:
        open(SHA, "w").write(text)
    print(f"oracle/_ref: {len(lines)} reference modules, manifest {SHA}")
    return True


if __name__ == "__main__":
    main()
