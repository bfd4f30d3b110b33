"""VAE-GAN step (config 2, batch 64, epoch 200: all six loss terms) replayed from its hipGraph, for rocprofv3 --stats."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oxford_102_flower_gan_vae_latent_diffusion_amd import modules as M, ops, steps
dev = torch.device("cuda:0")
torch.manual_seed(0)
ae = M.SimpleAutoencoder().to(dev).train()
D = M.Discriminator64().to(dev).train()
vgg = M.VGGPerceptualLoss(dev)  # seeded-synthetic weights (no network for the pretrained ones)
tr = steps.VAEGANTrainer(ae, D, vgg, lr=1e-4)
B = 64
data = torch.empty(B, 3, 64, 64, device=dev); ops.philox_fill(data, 5, mode=2)
labels = torch.empty(B, dtype=torch.int64, device=dev); ops.philox_randint(labels, 6, 102)
for _ in range(3):
    tr.step(data, labels, epoch=200, num_epochs=300)
tr.capture(data, labels, epoch=200, num_epochs=300)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = int(os.environ.get("N", "20"))
for _ in range(n):
    tr.replay()
torch.cuda.synchronize()
print("vaegan step: %.3f ms" % ((time.perf_counter() - t0) / n * 1e3))
