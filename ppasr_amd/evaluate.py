"""Batched evaluation harness: the decode half of ``PPASRTrainer.evaluate`` (ppasr/trainer.py:592-645) and
``__decoder_result`` (:330-352) on top of the HIP hot path.  The loss half needs the training graph (attention decoder,
CTC loss) and is out of scope; ``evaluate`` returns the mean CER / WER only.

With ``torch.distributed`` initialised, every rank evaluates the batches ``rank::world`` and the per-utterance error
rates are summed with one all-reduce (utterance data parallelism)."""
import torch

from ppasr_amd.decoders.ctc_greedy_decoder import greedy_decoder_batch
from ppasr_amd.utils.metrics import cer, labels_to_string, wer

__all__ = ["decoder_result", "evaluate"]


def decoder_result(outs, vocabulary, decoder="ctc_greedy", beam_search_decoder=None):
    """trainer.py:330-352: outs [B,T',V] (device tensor or numpy) -> list[str]; every one of the T' rows is decoded."""
    if decoder == "ctc_greedy" or beam_search_decoder is None:
        return greedy_decoder_batch(outs, vocabulary)
    return beam_search_decoder.decode_batch_beam_search_offline(probs_split=outs)


def evaluate(model, batches, vocab_list, decoder="ctc_greedy", metrics_type="cer", beam_search_decoder=None,
             display_result=False):
    """model: any ppasr_amd model with ``get_encoder_out(inputs, input_lens)``;
    batches: iterable of (inputs [B,T,F], labels [B,U] (-1 padded), input_lens [B], label_lens [B]) like the reference's
    test_loader.  -> mean error rate (float), -1 if there is nothing to score (trainer.py:643)."""
    dist = torch.distributed.is_available() and torch.distributed.is_initialized()
    rank = torch.distributed.get_rank() if dist else 0
    world = torch.distributed.get_world_size() if dist else 1
    eos = len(vocab_list) - 1
    total, count = 0.0, 0
    for batch_id, (inputs, labels, input_lens, _label_lens) in enumerate(batches):
        if batch_id % world != rank:
            continue
        outs = model.get_encoder_out(inputs, input_lens)
        out_strings = decoder_result(outs, vocab_list, decoder, beam_search_decoder)
        labels_str = labels_to_string(labels, vocab_list, eos=eos)
        for out_string, label in zip(out_strings, labels_str):
            err = wer(out_string, label) if metrics_type == "wer" else cer(out_string, label)
            total += err
            count += 1
            if display_result:
                print(f"pred: {out_string}\nlabel: {label}\n{metrics_type}: {round(err, 6)}")
    if dist:
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.distributed.get_backend() == "nccl" else "cpu"
        t = torch.tensor([total, float(count)], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t)
        total, count = float(t[0]), int(t[1])
    return total / count if count > 0 else -1
