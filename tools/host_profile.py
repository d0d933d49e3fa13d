"""Host-side cost of one Mamba (cfg5) training step: python tools/host_profile.py"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from language_interpolation_amd import train as T
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = T.load_config(os.path.join(root, "configs", "high_order_interpolation.yaml"),
                    ["net=mamba", "net.n_layer=2", "optimizer=adam", "batch_size=16", "data.max_features=16"])
data = T.make_data(cfg, T.synthetic_text(1 << 16))
module = T.prediction_net(cfg, device="cuda")
opt = module.configure_optimizers()
gen = torch.Generator().manual_seed(0)
def step():
    opt.zero_grad(set_to_none=True)
    loss, _ = module.eval_step(data.sample(16, gen))
    loss.backward()
    opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20): step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
