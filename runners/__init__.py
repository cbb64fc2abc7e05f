"""Drop-in `runners` package: same module paths and class names the reference's drivers import
(/root/reference/eval_sde_adv.py:27-31, eval_sde_adv_bpda.py), backed by the MI355X engine."""
