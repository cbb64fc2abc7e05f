"""Put the reference's UNCHANGED driver where the GPU box can load it (build container only: needs /root/reference).

    python tests/golden/make_ref_driver_fixture.py        (also run by __graft_entry__.build() when the reference is present)

/root/reference does not exist on the GPU box, and reference sources are never committed to this repository.  The copy therefore
goes to tests/golden/_ref_driver/ - listed in .gitignore (stays out of history) but not in .gpurunignore (travels with the
snapshot, like the built .so files) - and only its sha256 is tracked (tests/golden/ref_driver.sha256), so that
tests/test_reference_driver.py can tell on the GPU box that the file it loads is byte for byte the reference's eval_sde_adv.py."""
import hashlib
import os
import shutil

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/eval_sde_adv.py"
DST_DIR = os.path.join(HERE, "_ref_driver")
DST = os.path.join(DST_DIR, "eval_sde_adv.py")
SHA = os.path.join(HERE, "ref_driver.sha256")


def main():
    if not os.path.exists(SRC):
        print("reference checkout not present: fixture not (re)generated")
        return False
    os.makedirs(DST_DIR, exist_ok=True)
    shutil.copyfile(SRC, DST)
    digest = hashlib.sha256(open(DST, "rb").read()).hexdigest()
    line = f"{digest}  eval_sde_adv.py\n"
    if not os.path.exists(SHA) or open(SHA).read() != line:
        open(SHA, "w").write(line)
    print("fixture:", DST, digest)
    return True


if __name__ == "__main__":
    main()
