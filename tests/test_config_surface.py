"""CPU: the reference's config surface (SURVEY.md section 8b "Config surface"): the yml keys of configs/*.yml over the defaults
of config.py:7-56, the merge checks of config.py:61-87, the switches of train.py:372-375 / validate.py:240-241, the model
kwargs of train.py:58-66.  The three yml SHAPES below are committed key lists with the values the reference's files hold
(configs/svqa_DualVGR_20.yml, msvd_qa_DualVGR.yml, msrvtt_qa_DualVGR_16.yml) -- data, not the files."""
import argparse

import pytest
import yaml

from dualvgr_videoqa_amd import config as C

# (exp_name, num_of_nodes, dataset.name, data_dir) of the reference's three files; everything else is common to them
VARIANTS = {"svqa": ("expSVQA", 20, "svqa", "data/svqa"), "msvd": ("expMSVD-QA", 8, "msvd-qa", "data/msvd-qa"),
            "msrvtt": ("expMSRVTT-QA", 16, "msrvtt-qa", "data/msrvtt-qa")}


def yml_shape(exp, nodes, name, data_dir):
    return {"gpu_id": 1, "multi_gpus": False, "num_workers": 4, "seed": 666, "exp_name": exp, "model_type": "DualVGR",
            "graph_module": "GAT", "graph_layers": 1,
            "train": {"lr": 0.0001, "batch_size": 256, "restore": False, "max_epochs": 25, "word_dim": 300, "module_dim": 768,
                      "glove": True, "num_of_nodes": nodes},
            "val": {"flag": True}, "test": {"test_num": 0, "write_preds": False},
            "dataset": {"name": name, "data_dir": data_dir, "save_dir": "results/"}}


@pytest.mark.parametrize("which", sorted(VARIANTS))
def test_reference_yml_shapes_parse(which, tmp_path):
    exp, nodes, name, data_dir = VARIANTS[which]
    f = tmp_path / "c.yml"
    f.write_text(yaml.safe_dump(yml_shape(exp, nodes, name, data_dir)))
    cfg = C.cfg_from_file(str(f))
    assert cfg.train.num_of_nodes == nodes and cfg.dataset.name == name and cfg.graph_module == "GAT" and cfg.model_type == "DualVGR"
    assert cfg.train.vision_dim == 2048 and cfg.train.train_num == 0            # defaults survive (config.py:24,28)
    assert cfg.val.val_num == 0 and cfg.exp_name == exp and cfg.train.batch_size == 256
    mk = C.model_kwargs(cfg, unit_layers=3, vocab={"x": 1})
    assert mk == dict(vision_dim=2048, module_dim=768, word_dim=300, vocab={"x": 1}, num_of_nodes=nodes, graph_module="GAT",
                      graph_layers=1, unit_layers=3)
    p = C.dataset_paths(cfg, "val")
    assert p["question_pt"] == f"{data_dir}/{name}_val_questions.pt" and p["appearance_feat"] == f"{data_dir}/{name}_appearance_feat.h5"
    assert p["vocab_json"] == f"{data_dir}/{name}_vocab.json" and p["motion_feat"] == f"{data_dir}/{name}_motion_feat.h5"


def test_defaults_are_the_reference_defaults():
    d = C.defaults()
    assert (d.gpu_id, d.num_workers, d.multi_gpus, d.seed, d.model_type, d.graph_module, d.graph_layers) == (0, 4, False, 666, "baseline", "GCN", 1)
    assert d.train == {"restore": False, "lr": 0.0001, "batch_size": 32, "max_epochs": 25, "vision_dim": 2048, "word_dim": 300,
                       "module_dim": 768, "train_num": 0, "glove": True, "num_of_nodes": 8}
    assert d.test == {"test_num": 0, "write_preds": False} and d.exp_name == "defaultExp"
    assert C.defaults() is not d and C.defaults().train is not d.train           # fresh trees


def test_merge_errors_as_the_reference():
    with pytest.raises(KeyError, match="not a valid config key"):
        C.cfg_from_text("bogus_key: 1")
    with pytest.raises(KeyError, match="train.nodes is not a valid config key"):
        C.cfg_from_text("train:\n  nodes: 3")
    with pytest.raises(ValueError, match="Type mismatch"):
        C.cfg_from_text("train:\n  lr: 'fast'")
    with pytest.raises(ValueError, match="Type mismatch"):
        C.cfg_from_text("seed: 1.5")
    with pytest.raises(ValueError, match="Type mismatch"):
        C.cfg_from_text("train: 3")
    assert C.cfg_from_text("").seed == 666                                        # an empty file leaves the defaults


def test_cli_switches_and_their_defaults(tmp_path):
    ap = C.add_cli(argparse.ArgumentParser())
    a = ap.parse_args([])
    assert (a.cfg_file, a.alpha, a.beta, a.unit_layers) == (None, 1, 1e-8, 1)     # train.py:372-375
    a = ap.parse_args(["--alpha", "0.5", "--beta", "1e-6", "--unit_layers", "3"])
    assert (a.alpha, a.beta, a.unit_layers) == (0.5, 1e-6, 3)
    v = C.add_cli(argparse.ArgumentParser(), validate=True).parse_args(["--unit_layers", "2"])
    assert v.unit_layers == 2 and not hasattr(v, "alpha")                         # validate.py:240-241
    f = tmp_path / "c.yml"
    f.write_text(yaml.safe_dump(yml_shape(*VARIANTS["msvd"])))
    cfg, args = C.parse(["--cfg", str(f), "--unit_layers", "3"])
    assert cfg.train.num_of_nodes == 8 and args.unit_layers == 3
    assert C.step_kwargs(cfg, args.alpha, args.beta) == dict(lr=0.0001, alpha=1.0, beta=1e-8, max_norm=12.0)
    f.write_text("dataset:\n  name: 'tgif'\n")
    with pytest.raises(AssertionError):
        C.parse(["--cfg", str(f)])                                                # train.py:381
