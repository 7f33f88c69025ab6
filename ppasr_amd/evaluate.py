"""Batched evaluation harness: the decode half of ``PPASRTrainer.evaluate`` (ppasr/trainer.py:592-645) and
``__decoder_result`` (:330-352) on top of the HIP hot path.  The loss half needs the training graph (attention decoder,
CTC loss) and is out of scope; ``evaluate`` returns the mean CER / WER only.

With ``torch.distributed`` initialised, every rank evaluates the batches ``rank::world`` and the per-utterance error
rates are summed with one all-reduce (utterance data parallelism, NOTES.md §6).  For variable-length batches
``shard="buckets"`` splits EVERY batch over the ranks instead (``parallel.decode_ragged``: whole length buckets per rank
by padded work, one all-gather of the hypotheses per batch), so that ranks finish together whatever the batch order."""
import torch

from ppasr_amd.decoders.beam_search_decoder import beam_search_ids
from ppasr_amd.decoders.ctc_greedy_decoder import greedy_decode_ids, greedy_decoder_batch
from ppasr_amd.parallel import set_skip_padding_if_built
from ppasr_amd.utils.metrics import cer, labels_to_string, wer

__all__ = ["decoder_result", "evaluate"]


def _ids_to_text(ids, vocabulary, greedy):
    """Token ids -> string the way the reference's two decoders do it: greedy_decoder replaces "<space>" by " "
    (ctc_greedy_decoder.py:31); the beam-search module joins the vocabulary entries as they are (its vocabulary is handed
    over with " " already in place, beam_search_decoder.py:59-73)."""
    text = "".join(vocabulary[i] for i in ids)
    return text.replace("<space>", " ") if greedy else text


def _use_greedy(decoder, beam_search_decoder):
    return decoder == "ctc_greedy" or beam_search_decoder is None


def _reset_scorer(beam_search_decoder):
    """decode_batch_beam_search_offline re-applies the decoder object's alpha / beta to the scorer on every call
    (beam_search_decoder.py:60-61); every beam route of this module does the same, so alpha / beta tuned on the decoder
    object after construction are honoured whichever route decodes."""
    sc = getattr(beam_search_decoder, "_ext_scorer", None)
    if sc is not None:
        sc.reset_params(beam_search_decoder.alpha, beam_search_decoder.beta)


def decoder_result(outs, vocabulary, decoder="ctc_greedy", beam_search_decoder=None, frame_lens=None):
    """trainer.py:330-352: outs [B,T',V] (device tensor or numpy) -> list[str]; every one of the T' rows is decoded
    (the reference's behaviour) unless ``frame_lens`` [B] names the valid frames of every utterance."""
    if frame_lens is None:
        if decoder == "ctc_greedy" or beam_search_decoder is None:
            return greedy_decoder_batch(outs, vocabulary)
        return beam_search_decoder.decode_batch_beam_search_offline(probs_split=outs)
    greedy = _use_greedy(decoder, beam_search_decoder)
    if greedy:
        tokens, n, _, _, _ = greedy_decode_ids(outs, frame_lens)
    else:
        d = beam_search_decoder
        _reset_scorer(d)
        tokens, n, _, _ = beam_search_ids(outs, d.beam_size, d.cutoff_prob, d.cutoff_top_n, d.blank_id,
                                          frame_lens=frame_lens, nbest=1, ext_scorer=d._ext_scorer)
        tokens, n = tokens[:, 0], n[:, 0]
    tk, nn = tokens.cpu(), n.cpu()
    return [_ids_to_text(tk[b, :max(int(nn[b]), 0)].tolist(), vocabulary, greedy) for b in range(tk.shape[0])]


def evaluate(model, batches, vocab_list, decoder="ctc_greedy", metrics_type="cer", beam_search_decoder=None,
             display_result=False, trim_padding=False, overlap_decode=True, shard="batches"):
    """model: any ppasr_amd model with ``get_encoder_out(inputs, input_lens)``;
    batches: iterable of (inputs [B,T,F], labels [B,U] (-1 padded), input_lens [B], label_lens [B]) like the reference's
    test_loader.  -> mean error rate (float), -1 if there is nothing to score (trainer.py:643).
    ``trim_padding=True`` (not the reference's behaviour, which decodes the padded rows of every utterance too): the
    encoder runs in its ragged-batch mode and the decoders stop at each utterance's last valid frame.
    ``overlap_decode``: encode batch i+1 on a second HIP stream while batch i is being decoded (same results).
    ``shard``: "batches" = batch i goes to rank i % world; "buckets" = every batch is cut into length buckets that are
    dealt to the ranks by padded work (ragged batches).  "buckets" ALWAYS has ``trim_padding=True`` semantics -- the
    encoder skips the padding and decoding covers each utterance's valid frames only, whatever ``trim_padding`` says --
    so for batches with padding its error rate equals ``shard="batches", trim_padding=True``, not the reference's
    padded-row decoding; with ``decoder != "ctc_greedy"`` a ``beam_search_decoder`` is required (no silent greedy)."""
    dist = torch.distributed.is_available() and torch.distributed.is_initialized()
    if shard == "buckets":
        return _evaluate_bucketed(model, batches, vocab_list, decoder, metrics_type, beam_search_decoder, display_result,
                                  torch.distributed if dist else None)
    if shard != "batches":
        raise ValueError(f"shard={shard!r}: 'batches' or 'buckets'")
    rank = torch.distributed.get_rank() if dist else 0
    world = torch.distributed.get_world_size() if dist else 1
    eos = len(vocab_list) - 1
    total, count = 0.0, 0
    dev = getattr(model, "device", None)
    pipelined = overlap_decode and dev is not None and torch.device(dev).type == "cuda"
    enc_stream = torch.cuda.Stream(device=dev) if pipelined else None

    def encode(inputs, input_lens):
        frame_lens = None
        if trim_padding:
            set_skip_padding_if_built(model, True)
            frame_lens = model.valid_out_frames(input_lens, inputs.shape[1])
        try:
            outs = model.get_encoder_out(inputs, input_lens)
        finally:
            if trim_padding:
                set_skip_padding_if_built(model, False)
        return outs, frame_lens

    def score(outs, frame_lens, labels):
        nonlocal total, count
        out_strings = decoder_result(outs, vocab_list, decoder, beam_search_decoder, frame_lens)
        labels_str = labels_to_string(labels, vocab_list, eos=eos)
        for out_string, label in zip(out_strings, labels_str):
            err = wer(out_string, label) if metrics_type == "wer" else cer(out_string, label)
            total += err
            count += 1
            if display_result:
                print(f"pred: {out_string}\nlabel: {label}\n{metrics_type}: {round(err, 6)}")

    # Two-stage pipeline on two HIP streams: the encoder of batch i+1 is queued on its own stream before the host waits
    # for the decode of batch i (beam search occupies one CU per utterance -- the rest of the chip encodes meanwhile).
    pending = None  # (outs, frame_lens, labels, event) of the batch whose decode has not run yet
    for batch_id, (inputs, labels, input_lens, _label_lens) in enumerate(batches):
        if batch_id % world != rank:
            continue
        if not pipelined:
            outs, frame_lens = encode(inputs, input_lens)
            score(outs, frame_lens, labels)
            continue
        dec_stream = torch.cuda.current_stream(dev)
        enc_stream.wait_stream(dec_stream)  # (inputs prepared on the caller's stream)
        with torch.cuda.stream(enc_stream):
            outs, frame_lens = encode(inputs, input_lens)
            ev = torch.cuda.Event()
            ev.record(enc_stream)
        if pending is not None:
            score(*pending[:3])
        dec_stream.wait_event(ev)
        outs.record_stream(dec_stream)
        if frame_lens is not None:
            frame_lens.record_stream(dec_stream)
        pending = (outs, frame_lens, labels)
    if pending is not None:
        score(*pending[:3])
    if dist:
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.distributed.get_backend() == "nccl" else "cpu"
        t = torch.tensor([total, float(count)], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t)
        total, count = float(t[0]), int(t[1])
    return total / count if count > 0 else -1


def _evaluate_bucketed(model, batches, vocab_list, decoder, metrics_type, beam_search_decoder, display_result, dist):
    """Every rank walks every batch and decodes ITS length buckets of it; after the gather each rank holds all
    hypotheses of the batch, so the error sums are identical on every rank and need no further collective."""
    from ppasr_amd.parallel import beam_ids_decoder, decode_ragged, greedy_ids_decoder
    if decoder != "ctc_greedy" and beam_search_decoder is None:
        raise ValueError(f"evaluate(shard='buckets', decoder={decoder!r}) needs a beam_search_decoder")
    greedy = _use_greedy(decoder, beam_search_decoder)
    if greedy:
        dec = greedy_ids_decoder()
    else:
        d = beam_search_decoder
        _reset_scorer(d)
        dec = beam_ids_decoder(d.beam_size, d.cutoff_prob, d.cutoff_top_n, d.blank_id, d._ext_scorer)
    eos = len(vocab_list) - 1
    total, count = 0.0, 0
    for inputs, labels, input_lens, _label_lens in batches:
        lens = [int(v) for v in torch.as_tensor(input_lens).tolist()]
        tokens, n, _ = decode_ragged(model, inputs, lens, dec, dist=dist)
        tk, nn = tokens.cpu(), n.cpu()
        labels_str = labels_to_string(labels, vocab_list, eos=eos)
        for b, label in enumerate(labels_str):
            out_string = _ids_to_text(tk[b, :max(int(nn[b]), 0)].tolist(), vocab_list, greedy)
            err = wer(out_string, label) if metrics_type == "wer" else cer(out_string, label)
            total += err
            count += 1
            if display_result:
                print(f"pred: {out_string}\nlabel: {label}\n{metrics_type}: {round(err, 6)}")
    return total / count if count > 0 else -1
