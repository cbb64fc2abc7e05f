"""Put the reference's UNCHANGED driver where the GPU box can load it (build container only: needs /root/reference).

    python tests/golden/make_ref_driver_fixture.py        (also run by __graft_entry__.build() when the reference is present)

/root/reference does not exist on the GPU box, and reference sources are never committed to this repository.  The copy therefore
goes to tests/golden/_ref_driver/ - listed in .gitignore (stays out of history) but not in .gpurunignore (travels with the
snapshot, like the built .so files) - and only its sha256 is tracked (tests/golden/ref_driver.sha256), so that
tests/test_reference_driver.py can tell on the GPU box that the files it loads are byte for byte the reference's."""
import hashlib
import os
import shutil

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
# (source under /root/reference, name under tests/golden/_ref_driver/): the two callers of SURVEY 8(b) - eval_sde_adv.py and, round 6,
# eval_sde_adv_bpda.py with the attack class it drives the model through (bpda_eot/bpda_eot_attack.py: pure torch)
FILES = [("eval_sde_adv.py", "eval_sde_adv.py"), ("eval_sde_adv_bpda.py", "eval_sde_adv_bpda.py"),
         ("bpda_eot/bpda_eot_attack.py", "bpda_eot_attack.py")]
SRC = os.path.join(REF, FILES[0][0])
DST_DIR = os.path.join(HERE, "_ref_driver")
DST = os.path.join(DST_DIR, "eval_sde_adv.py")
SHA = os.path.join(HERE, "ref_driver.sha256")


def main():
    if not os.path.exists(SRC):
        print("reference checkout not present: fixture not (re)generated")
        return False
    os.makedirs(DST_DIR, exist_ok=True)
    lines = []
    for src, name in FILES:
        dst = os.path.join(DST_DIR, name)
        shutil.copyfile(os.path.join(REF, src), dst)
        digest = hashlib.sha256(open(dst, "rb").read()).hexdigest()
        lines.append(f"{digest}  {name}\n")
        print("fixture:", dst, digest)
    text = "".join(lines)
    if not os.path.exists(SHA) or open(SHA).read() != text:
        open(SHA, "w").write(text)
    return True


if __name__ == "__main__":
    main()
