"""TEST INFRASTRUCTURE ONLY -- never imported by the product package.

CPU restatement of the reference's epoch-loop protocol (best-model tracking, early stop, results JSON).  The
reference keeps this logic inline in ``train()`` / ``main()`` (train.py:2355-2357, 2522-2543, 2744-2757,
4571-4668), so it cannot be called in isolation: this restatement is pinned by reading only ("parity unpinned"
by reference outputs), whereas the checkpoint file layout IS pinned by ``tests/golden/checkpoint_layout.json``,
frozen from the reference's own ``save_checkpoint`` / ``load_checkpoint`` by ``oracle/make_golden_host.py``.
"""
from __future__ import annotations


def early_stop_trace(val_losses, patience):
    """Walk train.py:2522-2543 + 2744-2757 over a sequence of (already rank-averaged) validation losses.
    Returns one (is_best, best_val_loss, best_epoch, patience_counter, should_stop) per epoch run."""
    best_val_loss, best_epoch, patience_counter = float("inf"), 0, 0
    out = []
    for epoch, val_loss in enumerate(val_losses, start=1):
        is_best = val_loss < best_val_loss
        if is_best:
            best_val_loss, best_epoch, patience_counter = val_loss, epoch, 0
        else:
            patience_counter += 1
        should_stop = patience_counter >= patience
        out.append((is_best, best_val_loss, best_epoch, patience_counter, should_stop))
        if should_stop:
            break
    return out


def results_payload(results, arch, fold):
    """train.py:4571-4668, statement by statement."""
    history = results.get("history", [])
    train_losses = [h.get("loss", 0.0) for h in history]
    val_losses = [h.get("val_loss", 0.0) for h in history]
    val_r2s = [h.get("val_r2", 0.0) for h in history]
    val_corrs = [h.get("val_corr", 0.0) for h in history]
    val_maes = [h.get("val_mae", 0.0) for h in history]
    best_epoch = results["best_epoch"]
    best_val_r2 = val_r2s[best_epoch - 1] if best_epoch > 0 and len(val_r2s) >= best_epoch else 0.0
    best_val_mae = val_maes[best_epoch - 1] if best_epoch > 0 and len(val_maes) >= best_epoch else 0.0
    best_val_corr = val_corrs[best_epoch - 1] if best_epoch > 0 and len(val_corrs) >= best_epoch else 0.0
    n_params = results.get("n_parameters", 0)
    per_session_r2, per_session_corr, per_session_loss = {}, {}, {}
    if history:
        best_entry = history[best_epoch - 1] if best_epoch <= len(history) else history[-1]
        for key, value in best_entry.items():
            if key.startswith("val_") and "_r2" in key and key != "val_r2":
                session = key[4:-3]
                if session:
                    per_session_r2[session] = value
            elif key.startswith("val_") and "_corr" in key and key != "val_corr":
                session = key[4:-5]
                if session:
                    per_session_corr[session] = value
            elif key.startswith("val_") and "_loss" in key and key != "val_loss":
                session = key[4:-5]
                if session:
                    per_session_loss[session] = value
    test_metrics = results.get("test_metrics", {})
    final_gate_mean = history[-1].get("gate_mean", None) if history else None
    return {
        "architecture": arch, "fold": fold if fold is not None else 0,
        "best_val_r2": best_val_r2, "best_val_mae": best_val_mae, "best_val_corr": best_val_corr,
        "best_val_loss": results["best_val_loss"], "best_epoch": best_epoch,
        "train_losses": train_losses, "val_losses": val_losses, "val_r2s": val_r2s, "val_corrs": val_corrs,
        "per_session_r2": per_session_r2, "per_session_corr": per_session_corr, "per_session_loss": per_session_loss,
        "per_session_test_results": test_metrics.get("per_session_test_results", []),
        "test_avg_r2": test_metrics.get("test_avg_r2") or test_metrics.get("r2"),
        "test_avg_corr": test_metrics.get("test_avg_corr") or test_metrics.get("corr"),
        "test_avg_delta": test_metrics.get("test_avg_delta") or test_metrics.get("loss"),
        "total_time": results.get("total_time", 0.0), "epochs_trained": len(history), "n_parameters": n_params,
        "completed_successfully": True, "gate_mean": final_gate_mean,
        "matched_corr": test_metrics.get("matched_corr"), "matched_r2": test_metrics.get("matched_r2"),
        "matched_nrmse": test_metrics.get("matched_nrmse"),
        "baseline_corr": test_metrics.get("baseline_corr"), "baseline_r2": test_metrics.get("baseline_r2"),
    }
