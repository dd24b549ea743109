#!/usr/bin/env python3
"""Config 1 of BASELINE.json on the MI355X package: the reference's demo experiment
(/root/reference/examples/demo.py: 300 synthetic 84-ROI subjects, Watts-Strogatz k=8 beta=0.15,
70/15/15 split, GCNConnectome and GraphSAGEConnectome hidden 64, batch 16, Adam lr 1e-3 wd 1e-4,
30 epochs, patience 8) with nothing changed but the import root and the device string.

    python examples/demo.py [--device cuda] [--epochs 30] [--quiet]

The reference documents ~55-70 % test accuracy for both models (weak brain-behaviour
correlations); `tests/test_gpu_demo.py` checks this script's numbers against that band.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import connectome_gnn_amd as connectome_gnn  # noqa: E402  (the whole migration)
from connectome_gnn_amd.synthetic import small_world_stats  # noqa: E402


def banner(text: str) -> None:
    print("\n" + "=" * 60 + f"\n  {text}\n" + "=" * 60)


def run(device: str = "cuda", epochs: int = 30, verbose: bool = True, subjects: int = 300,
        regions: int = 84, batch_size: int = 16, hidden: int = 64) -> dict:
    torch.manual_seed(42)
    say = print if verbose else (lambda *a, **k: None)
    if verbose:
        banner("1. Synthetic connectome dataset")
    graphs = connectome_gnn.generate_dataset(num_subjects=subjects, num_regions=regions, k=8, beta=0.15,
                                             trait_idx=0, seed=42)
    g0 = graphs[0]
    say(f"  {subjects} subjects x {regions} regions; nodes={g0.num_nodes} edges={g0.num_edges} "
        f"features/node={g0.num_features}")
    sw = small_world_stats(graphs[:20])
    say(f"  small-world check (20 subjects): clustering {sw['mean_clustering']:.3f}, "
        f"path length {sw['mean_avg_path_length']:.3f}")
    ones = sum(int(g.label) for g in graphs)
    say(f"  labels: class 0 = {subjects - ones}, class 1 = {ones}")

    n_train, n_val = int(0.7 * subjects), int(0.15 * subjects)
    parts = (graphs[:n_train], graphs[n_train:n_train + n_val], graphs[n_train + n_val:])
    loaders = [connectome_gnn.ConnectomeDataLoader(p, batch_size=batch_size, shuffle=(i == 0))
               for i, p in enumerate(parts)]
    say(f"  split: train {len(parts[0])} | val {len(parts[1])} | test {len(parts[2])}")

    results = {}
    for title, cls in (("GCNConnectome", connectome_gnn.GCNConnectome),
                       ("GraphSAGEConnectome", connectome_gnn.GraphSAGEConnectome)):
        if verbose:
            banner(f"Training {title}")
        model = cls(in_channels=g0.num_features, hidden_dim=hidden, num_classes=2, num_layers=3, dropout=0.3)
        params = sum(p.numel() for p in model.parameters())
        say(f"  parameters: {params:,}")
        trainer = connectome_gnn.Trainer(
            model, torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4), device=device)
        history = trainer.fit(loaders[0], loaders[1], num_epochs=epochs, patience=8, verbose=verbose)
        test = trainer.evaluate(loaders[2])
        say(f"\n  {title} test accuracy: {test['accuracy']:.3f} ({test['correct']}/{test['total']})")
        results[title] = {"params": params, "test": test, "history": history, "impl": model.impl_used}

    if verbose:
        banner("Summary")
        for title, r in results.items():
            print(f"  {title:<22} test acc {r['test']['accuracy']:.3f}   best val loss "
                  f"{min(r['history']['val_loss']):.4f}   path: {r['impl']}")
        print("  (the reference reports ~55-70 % for this experiment)")
    return results


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--epochs", type=int, default=30)
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args()
    run(a.device, a.epochs, not a.quiet)
