"""CPU tests of the training entry point's boundary: the reference launcher's full flag list is accepted
(configs/job_configs/training/filtered/*_sn.sh:65-100 -> Fire(main), scripts/training/train_timestamps.py:2098-2134), the
checkpoint pieces this implementation writes load into the objects the REFERENCE's load_ckpt builds (:1031-1056), and a
reference-pickled ``dims`` loads here without the reference package."""
import importlib.util
import os
import pickle
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tt():
    spec = importlib.util.spec_from_file_location("tt_cli", os.path.join(ROOT, "scripts", "training", "train_timestamps.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


# the torchrun line of text_heurs_seg_edit_dist_0.7_edit_dist_0.5_sn.sh:65-100 with its shell variables filled in
LAUNCHER = ["--model_variant=medium", "--exp_name=exp_evalbs32_092525", "--job_type=train", "--samples_dicts_dir=/data/samples",
            "--train_steps=1048576", "--epoch_steps=16384", "--ckpt_file_name=None", "--ckpt_dir=checkpoints", "--log_dir=logs",
            "--eval_dir=data/eval", "--run_id_dir=run_ids", "--lr=1.5e-3", "--betas=(0.9, 0.98)", "--eps=1e-6", "--weight_decay=0.1",
            "--max_grad_norm=1.0", "--eff_batch_size=2048", "--train_batch_size=32", "--eval_batch_size=32", "--num_workers=10",
            "--prefetch_factor=2", "--pin_memory=True", "--shuffle=True", "--persistent_workers=True", "--run_eval=False",
            "--train_log_freq=20000", "--eval_freq=20000", "--ckpt_freq=2500", "--verbose=False", "--precision=bfloat16",
            "--hardware=H100", "--async_eval=False", "--eval_script_path=scripts/eval/eval.py", "--eval_wandb_log=False",
            "--eval_on_gpu=True"]


def test_reference_launcher_flags_are_accepted(tt, capsys):
    a = tt.parse_args(LAUNCHER)
    assert a.model_variant == "medium" and a.ckpt_file_name == "" and a.betas == (0.9, 0.98) and a.lr == 1.5e-3
    assert a.eff_batch_size == 2048 and a.train_batch_size == 32 and a.pin_memory is True and a.run_eval is False
    assert a.precision == "bfloat16" and a.train_steps == 1048576
    assert "ignored_flags" in capsys.readouterr().out  # samples_dicts_dir etc. are reported, not silently dropped
    # space-separated form and reference defaults
    b = tt.parse_args(["--model_variant", "tiny", "--betas", "(0.8, 0.9)"])
    assert b.betas == (0.8, 0.9) and b.eps == 1e-6 and b.weight_decay == 0.1 and b.max_grad_norm == 1.0 and b.eff_batch_size == 256
    assert b.train_batch_size == 8 and b.num_workers == 10 and b.ckpt_freq == 2500 and b.train_log_freq == 20000
    assert tt.accumulation_steps(2048, 8, 32) == 8 and tt.accumulation_steps(256, 64, 8) == 1


def test_precision_and_unknown_flags(tt):
    assert tt.parse_args(["--precision=float32"]).precision == "float32"
    with pytest.raises(SystemExit, match="float16"):
        tt.parse_args(["--precision=float16"])
    with pytest.raises(SystemExit, match="unknown flag"):
        tt.parse_args(["--no_such_flag=1"])


def test_multi_rank_without_a_rendezvous_fails_at_once(tt, monkeypatch):
    """WORLD_SIZE > 1 with no MASTER_ADDR / MASTER_PORT (srun, mp.spawn without the exports): a per-process default port would leave
    every rank waiting on a different rendezvous until the RCCL timeout -- the script must stop with a message instead."""
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.delenv("MASTER_ADDR", raising=False)
    monkeypatch.delenv("MASTER_PORT", raising=False)
    with pytest.raises(SystemExit) as e:
        tt.main(["--model_variant=tiny", "--train_steps=1"])
    assert "MASTER_ADDR and MASTER_PORT not set" in str(e.value)


def test_checkpoint_pieces_load_on_the_reference_side(tt, tmp_path):
    """What the reference's load_ckpt does with a checkpoint (train_timestamps.py:1031-1056), piece by piece, on OUR file:
    OLMoASR(dims=ckpt['dims']) with the UNMODIFIED reference model class, model.load_state_dict, AdamW.load_state_dict,
    LambdaLR.load_state_dict, gen_inf_ckpt's ``dims.__dict__``."""
    from oracle import model_oracle as mo
    dims = mo.Dims(80, 1500, 64, 1, 1, 51864, 448, 64, 1, 1)
    sd = mo.init_state_dict(dims, seed=0)
    ck = tt.build_checkpoint(dict(sd), {"state": {}, "param_groups": []}, {"scale": 65536.0, "_growth_tracker": 3, "growth_factor": 2.0,
                             "backoff_factor": 0.5, "growth_interval": 2000}, global_step=7, local_step=56, epoch=0, dims=dims,
                             lr=1.5e-3, train_steps=1000, cursor=5, optimizer_steps=6)
    path = tmp_path / "ck.pt"
    torch.save(ck, path)
    raw = torch.load(path, map_location="cpu", weights_only=False)  # plain torch.load: nothing of this package is needed
    assert isinstance(raw["dims"], types.SimpleNamespace) and raw["dims"].n_audio_state == 64 and raw["dims"].__dict__["n_vocab"] == 51864
    assert set(raw) >= {"global_step", "local_step", "epoch", "best_eval_wer", "model_state_dict", "optimizer_state_dict",
                        "scaler_state_dict", "scheduler_state_dict", "dims"}
    # LambdaLR of the reference's prepare_sched (:772-781) accepts the scheduler entry and continues from step 7
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=1.5e-3)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: tt.lr_lambda(s, 1000))
    sched.load_state_dict(raw["scheduler_state_dict"])
    assert sched.last_epoch == 7 and abs(sched.get_last_lr()[0] - 1.5e-3 * tt.lr_lambda(7, 1000)) < 1e-12
    opt.step()
    sched.step()
    assert sched.last_epoch == 8 and abs(opt.param_groups[0]["lr"] - 1.5e-3 * tt.lr_lambda(8, 1000)) < 1e-12
    scaler = torch.amp.GradScaler("cpu", enabled=True)
    scaler.load_state_dict(raw["scaler_state_dict"])
    # the unmodified reference model class builds from our dims object and loads our state_dict
    from oracle import ref_import
    if not ref_import.available():  # /root/reference is absent on the GPU box
        pytest.skip("reference tree not mounted")
    ref_model, _, _ = ref_import.load()
    model = ref_model.OLMoASR(dims=raw["dims"])
    res = model.load_state_dict(raw["model_state_dict"], strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    # AdamW state indices follow named_parameters() order: the committed fixture (oracle/gen_param_order.py) pins the
    # reference's order; tests/test_gpu_model.py checks the native model against the same file
    import json
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_param_order.json")))
    ref_tiny = ref_model.OLMoASR(dims=types.SimpleNamespace(**mo.VARIANTS["tiny"].__dict__))
    assert [n for n, _ in ref_tiny.named_parameters()] == want["tiny"]


def test_reference_pickled_dims_load_here(tmp_path):
    """A checkpoint written by the reference pickles ``olmoasr.config.model_dims.ModelDimensions``; hub.load_checkpoint
    resolves it when the reference package is not importable (the GPU box, a user's machine)."""
    import dataclasses

    from olmoasr_amd import hub
    from olmoasr_amd.config.model_dims import VARIANT_TO_DIMS
    names = ("olmoasr", "olmoasr.config", "olmoasr.config.model_dims")
    saved = {k: sys.modules.pop(k, None) for k in list(sys.modules) if k == "olmoasr" or k.startswith("olmoasr.")}
    ck_path = tmp_path / "refstyle.pt"
    try:
        for nm in names:  # a stand-in for the reference package, only while WRITING the file
            m = types.ModuleType(nm)
            m.__path__ = []
            sys.modules[nm] = m
        Fake = dataclasses.make_dataclass("ModelDimensions", [(f, int) for f in VARIANT_TO_DIMS["tiny"].__dataclass_fields__])
        Fake.__module__ = "olmoasr.config.model_dims"
        sys.modules["olmoasr.config.model_dims"].ModelDimensions = Fake
        torch.save({"dims": Fake(**VARIANT_TO_DIMS["base"].__dict__), "model_state_dict": {}}, ck_path)
        for nm in names:
            sys.modules.pop(nm, None)
        with pytest.raises(Exception):
            torch.load(ck_path, map_location="cpu", weights_only=False)  # not loadable without the module path ...
        ck = hub.load_checkpoint(str(ck_path))                            # ... but through the alias it is
        assert hub.dims_of(ck["dims"]) == VARIANT_TO_DIMS["base"]
        assert "olmoasr.config.model_dims" not in sys.modules             # and the alias does not leak
    finally:
        for nm in names:
            sys.modules.pop(nm, None)
        sys.modules.update(saved)


def test_load_model_by_name_without_network_fails_loudly_and_leaves_no_partial_file(tmp_path):
    """olmoasr.load_model("tiny") downloads into the cache (olmoasr/__init__.py:44-94); offline the error names the URL and nothing
    half-written stays behind."""
    import pytest
    from olmoasr_amd import hub
    with pytest.raises(RuntimeError, match="OLMoASR-tiny.en.pt"):
        hub.load_model("tiny", device="cpu", download_root=str(tmp_path))
    assert not list(tmp_path.glob("*.pt"))
    with pytest.raises(ValueError, match="Available models"):  # the reference's exception type and wording (olmoasr/__init__.py:135-138)
        hub.load_model("no-such-model", device="cpu")
