"""Developer probe: question-encoder fwd+bwd (B=160, T=26, d=512) with the MIOpen RNN vs torch's native LSTM cell path."""
import sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vqa_amd
from vqa_amd.modules import QuestionCoAttentionEncoder
from vqa_amd import train as T

dev = torch.device("cuda", 0)
b = T.synthetic_batch(160, (32, 32), 26, 10000, 1001, seed=1234)
image, question, label, lens = T.sort_batch(b["image"], b["question"], b["label"], b["ques_len"])
question = question.to(dev)
torch.manual_seed(0)
enc = QuestionCoAttentionEncoder(10000, 512, 512).to(dev)
outs = {}
for enabled in (True, False):
    with torch.backends.cudnn.flags(enabled=enabled):
        def step():
            for p in enc.parameters():
                p.grad = None
            w, p, s = enc(question, lens)
            (w.sum() + (p * p).sum() + (s * s).sum()).backward()
            return s
        for _ in range(5):
            s = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            s = step()
        torch.cuda.synchronize()
        print("miopen_rnn=%s  fwd+bwd %.3f ms" % (enabled, (time.perf_counter() - t0) * 50), flush=True)
        outs[enabled] = (s.detach().clone(), enc.sentence_lstm.weight_hh_l0.grad.clone())
print("max diff sentence", (outs[True][0] - outs[False][0]).abs().max().item(),
      "grad w_hh", (outs[True][1] - outs[False][1]).abs().max().item())
