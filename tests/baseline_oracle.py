"""Oracle results at BASELINE sizes, computed once per test process (the CPU oracle needs ~30-100 s per batch-32 model on the GPU box's host cores,
and five tests compare against the same three results).  Test infrastructure: only tests/ imports this."""
import functools

import numpy as np

BATCH = 32


@functools.lru_cache(maxsize=None)
def resnet_weights():
    from rten_amd.workloads import resnet50
    return resnet50.make_weights()


def resnet_input(seed=1234, batch=BATCH):
    """bench.py's rank-`seed - 1234` batch: U[0, 1) f32, 224 x 224."""
    return np.random.default_rng(seed).random((batch, 3, 224, 224), dtype=np.float32)


@functools.lru_cache(maxsize=None)
def resnet50_f32_logits(seed=1234, batch=BATCH):
    from oracle import models as om
    from rten_amd.workloads import resnet50
    return om.resnet50_forward(resnet50.conv_specs(), resnet_weights(), resnet_input(seed, batch))


@functools.lru_cache(maxsize=None)
def resnet50_int8_logits(seed=1234, batch=BATCH):
    from oracle import models as om
    from rten_amd.workloads import resnet50
    return om.resnet50_int8_forward(resnet50.conv_specs(), om.quantize_weights_int8(resnet_weights()), resnet_input(seed, batch))


@functools.lru_cache(maxsize=None)
def bert_base_case(batch=BATCH, seq=128):
    """(cfg, weights, ids, attention mask (ragged), token types, oracle last_hidden_state) of BASELINE configs[3]: 12 layers, hidden 768, 12 heads."""
    from oracle import models as om
    from rten_amd.workloads import bert
    cfg = bert.BertConfig(hidden=768, heads=12, layers=12, ffn=3072, vocab=4000, max_pos=seq)
    w = bert.make_weights(cfg)
    rng = np.random.default_rng(11)
    ids = rng.integers(0, cfg.vocab, (batch, seq))
    tts = rng.integers(0, 2, (batch, seq))
    am = np.ones((batch, seq), np.float32)
    for b in range(0, batch, 3):
        am[b, seq - 1 - 5 * (b % 7):] = 0  # padded tails of different lengths
    want = om.bert_forward(cfg, w, ids, am, tts)
    return cfg, w, ids, am, tts, want
