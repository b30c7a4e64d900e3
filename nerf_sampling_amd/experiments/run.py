"""Counterpart of nerf_sampling/experiments/run.py: train the DepthNet against a frozen pretrained NeRF.

    python -m nerf_sampling_amd.experiments.run -d lego [--iters 100000]

Same flags and overrides as the reference (run.py:16-113); wandb is out of scope (-w accepted, ignored).
"""

import os

import click
import torch
import yaml

from nerf_sampling_amd import ops
from nerf_sampling_amd.utils import load_obj_from_config, override_config, set_global_device

ROOT_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@click.command()
@click.option("-c", "--config", help="Path to configuration file.", type=str,
              default=f"{ROOT_DIR}/experiments/configs/lego.yaml", show_default=True)
@click.option("-dp", "--dataset_path", help="Path to dataset folder.", type=str, show_default=True)
@click.option("-d", "--dataset", help="Name of the dataset to train on.", type=str, show_default=True)
@click.option("-m", "--model", help="Model type.", type=str, default="lego_depth_net_module", show_default=True)
@click.option("-w", "--wandb", type=click.Choice(["online", "offline", "disabled"], case_sensitive=False),
              default="disabled", help="Ignored (wandb logging is out of scope).", show_default=True)
@click.option("-si", "--single_image", is_flag=True, default=False, help="Train sampling network on single image.")
@click.option("-sr", "--single_ray", is_flag=True, default=False, help="Train sampling network on single ray.")
@click.option("-ip", "--i_print", default=1000, help="Frequency of log printing.", show_default=True)
@click.option("--iters", default=100_000, show_default=True, help="Training iterations (EPOCHS in the reference).")
@click.option("--dtype", default="f32", type=click.Choice(["bf16", "f16", "f32", "f16x3"]), show_default=True,
              help="MFMA operand precision of the frozen-NeRF forward kernels (not in the reference).")
@click.option("--root", default=os.getcwd(), show_default=True, help="Directory holding dataset/ pretrained/ logs/.")
def main(**kw):
    """Run sampling-network training with the provided configuration (reference flow: run.py:79-155)."""
    with open(kw["config"], "r") as fin:
        config = yaml.safe_load(fin)[kw["model"]]
    k = config["kwargs"]
    k.update(single_image=kw["single_image"], single_ray=kw["single_ray"], i_print=kw["i_print"])
    root, dataset_name = kw["root"], kw["dataset"]
    datadir, ft_path = kw["dataset_path"], None
    if dataset_name is not None:
        datadir = f"{root}/dataset/{dataset_name}"
        ft_path = f"{root}/pretrained/nerf/{dataset_name}/200000.tar"
    if datadir is None:
        print("Please specify the name of the dataset or provide the path to the folder")
        return
    override_config(config=k, update={"depth_net_lr": 1e-4, "n_layers": 10, "layer_width": 256,
                                      "train_depth_net_only": True, "sphere_radius": 2})
    torch.manual_seed(42)
    set_global_device(k["device"])
    ops.set_compute_dtype(kw["dtype"])
    k.update(ft_path=ft_path, depth_net_path=None, datadir=datadir, basedir=f"{root}/logs")
    trainer = load_obj_from_config(cfg=config)
    trainer.train(N_iters=kw["iters"] + 1)


if __name__ == "__main__":
    main()
