import sys, traceback, torch
sys.path.insert(0,'/root/repo')
import jdet_amd.models
from jdet_amd.config.named import ORCNN_CFG
from jdet_amd.runner import Runner, synthetic_batch
dev=torch.device('cuda:0')
TARGET={3072,33554432}
def report(nbytes, what):
    if nbytes in TARGET:
        print("==", what, nbytes); print("".join(traceback.format_stack(limit=8)[:-2]))
oz=torch.zeros
def zeros(*a, **k):
    t=oz(*a, **k); report(t.numel()*t.element_size(), "torch.zeros"); return t
torch.zeros=zeros
ozl=torch.zeros_like
def zeros_like(*a, **k):
    t=ozl(*a, **k); report(t.numel()*t.element_size(), "zeros_like"); return t
torch.zeros_like=zeros_like
oz_=torch.Tensor.zero_
def zero_(self):
    report(self.numel()*self.element_size(), "zero_"); return oz_(self)
torch.Tensor.zero_=zero_
onz=torch.Tensor.new_zeros
def new_zeros(self,*a,**k):
    t=onz(self,*a,**k); report(t.numel()*t.element_size(),"new_zeros"); return t
torch.Tensor.new_zeros=new_zeros
torch.manual_seed(0)
runner = Runner(ORCNN_CFG, device=dev, graph=False)
images, targets = synthetic_batch(2, 512, dev, seed=3)
images = images.contiguous(memory_format=torch.channels_last)
runner.train_step(images, targets)
print("done")
