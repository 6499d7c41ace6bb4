"""The reference's config surface for the hot path's harness: the yml keys of `configs/*.yml` / `config.py:7-56` and the
command-line switches `--cfg --alpha --beta --unit_layers` (`train.py:372-375`; `validate.py:240-241` takes `--cfg
--unit_layers`).  A yml file written for the reference is read unchanged; what the reference does with a bad file it
does here too: an unknown key raises `KeyError('<key> is not a valid config key')`, a value whose type differs from the
default's raises `ValueError` (`config.py:61-87`).  `yaml.safe_load` instead of the reference's `yaml.load` without a
loader (an error on current PyYAML); plain dicts with attribute access instead of `easydict` (absent here).

Not a port of the reference's CLI or training script (out of scope) -- only the surface, so that
`bench.py --cfg`, `tools/train_synthetic.py --cfg` and a host that keeps the reference's own `train.py`
agree on names, defaults and meaning."""
import argparse
import copy

DATASETS = ("svqa", "msrvtt-qa", "msvd-qa")          # train.py:381


class Cfg(dict):
    """dict whose keys read as attributes (`cfg.train.lr`), recursively."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self[k] = v


def defaults():
    """A fresh tree of the reference's defaults (`config.py:7-56`)."""
    return _wrap({
        "gpu_id": 0, "num_workers": 4, "multi_gpus": False, "seed": 666, "model_type": "baseline", "graph_module": "GCN",
        "graph_layers": 1,
        "train": {"restore": False, "lr": 0.0001, "batch_size": 32, "max_epochs": 25, "vision_dim": 2048, "word_dim": 300,
                  "module_dim": 768, "train_num": 0, "glove": True, "num_of_nodes": 8},
        "val": {"flag": True, "val_num": 0},
        "test": {"test_num": 0, "write_preds": False},
        "dataset": {"name": "svqa", "data_dir": "", "appearance_feat": "{}_{}_appearance_feat.h5",
                    "motion_feat": "{}_{}_motion_feat.h5", "vocab_json": "{}_{}_vocab.json",
                    "train_question_pt": "{}_{}_train_questions.pt", "val_question_pt": "{}_{}_val_questions.pt",
                    "test_question_pt": "{}_{}_test_questions.pt", "save_dir": ""},
        "exp_name": "defaultExp",
    })


def _wrap(d):
    return Cfg({k: _wrap(v) if isinstance(v, dict) else v for k, v in d.items()})


def merge(loaded, cfg, _path=""):
    """Values of `loaded` (a parsed yml mapping) into `cfg`, with the reference's checks (`config.py:61-87`)."""
    if not isinstance(loaded, dict):
        return cfg
    for k, v in loaded.items():
        if k not in cfg:
            raise KeyError("{} is not a valid config key".format(_path + k))
        have = cfg[k]
        if isinstance(have, dict):
            if not isinstance(v, dict):
                raise ValueError("Type mismatch ({} vs. {}) for config key: {}".format(dict, type(v), _path + k))
            merge(v, have, _path + k + ".")
            continue
        if type(have) is not type(v):
            if isinstance(have, list) and isinstance(v, str):
                v = [int(x) for x in v.split(",")]
            elif have is None:
                if v == "None":
                    continue
            else:
                raise ValueError("Type mismatch ({} vs. {}) for config key: {}".format(type(have), type(v), _path + k))
        cfg[k] = v
    return cfg


def cfg_from_file(path, cfg=None):
    """Read a reference yml (`configs/svqa_DualVGR_20.yml` and its siblings) over the defaults."""
    import yaml
    with open(path, "r") as f:
        loaded = yaml.safe_load(f)
    return merge(loaded or {}, cfg if cfg is not None else defaults())


def cfg_from_text(text, cfg=None):
    import yaml
    return merge(yaml.safe_load(text) or {}, cfg if cfg is not None else defaults())


def add_cli(parser, validate=False):
    """The reference's switches with its defaults: `--cfg --alpha --beta --unit_layers` (`train.py:372-375`), or only
    `--cfg --unit_layers` for a validation script (`validate.py:240-241`).  `--cfg` defaults to None here (the reference
    defaults to a file name relative to its own working directory)."""
    parser.add_argument("--cfg", dest="cfg_file", help="optional config file", default=None, type=str)
    if not validate:
        parser.add_argument("--alpha", dest="alpha", help="optional loss parameter", default=1, type=float)
        parser.add_argument("--beta", dest="beta", help="optional loss parameter", default=1e-8, type=float)
    parser.add_argument("--unit_layers", dest="unit_layers", help="unit layers", default=1, type=int)
    return parser


def from_args(args):
    """cfg for parsed switches (the yml named by `--cfg` over the defaults); checks the dataset name as `train.py:381` does."""
    cfg = cfg_from_file(args.cfg_file) if getattr(args, "cfg_file", None) else defaults()
    if cfg.dataset.name not in DATASETS:
        raise AssertionError("dataset.name must be one of {}".format(DATASETS))
    return cfg


def model_kwargs(cfg, unit_layers=1, vocab=None):
    """Constructor arguments of `DualVGR` as `train.py:58-66` builds them from cfg -- plus `unit_layers`, which the reference
    parses and then drops in training (deliberate deviation: honoured here)."""
    t = cfg.train
    return dict(vision_dim=t["vision_dim"], module_dim=t["module_dim"], word_dim=t["word_dim"], vocab=vocab,
                num_of_nodes=t["num_of_nodes"], graph_module=cfg.graph_module, graph_layers=cfg.graph_layers,
                unit_layers=unit_layers)


def step_kwargs(cfg, alpha=1.0, beta=1e-8):
    """Arguments of `trainer.TrainStep` from cfg and the loss switches (`train.py:85,146-159`: Adam(lr), clip 12)."""
    return dict(lr=cfg.train["lr"], alpha=float(alpha), beta=float(beta), max_norm=12.0)


def dataset_paths(cfg, split="train"):
    """File names as `train.py:411-422` / `validate.py` form them: the script resets the templates to the one-placeholder
    forms `{}_appearance_feat.h5`, `{}_motion_feat.h5`, `{}_vocab.json`, `{}_<split>_questions.pt`, fills in `dataset.name`
    and joins with `dataset.data_dir` (the two-placeholder defaults of `config.py:44-49` are never used as they stand)."""
    import os
    d, name = cfg.dataset, cfg.dataset["name"]
    at = lambda tmpl: os.path.join(d["data_dir"], tmpl.format(name))
    return {"appearance_feat": at("{}_appearance_feat.h5"), "motion_feat": at("{}_motion_feat.h5"), "vocab_json": at("{}_vocab.json"),
            "question_pt": at("{}_" + split + "_questions.pt")}


def parse(argv=None, validate=False, parser=None):
    """`(cfg, args)` for a script that wants exactly the reference's switches (plus whatever `parser` already holds)."""
    ap = add_cli(parser or argparse.ArgumentParser(), validate=validate)
    args = ap.parse_args(argv)
    return from_args(args), args


def snapshot(cfg):
    return copy.deepcopy(dict(cfg))
