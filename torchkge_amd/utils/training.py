# -*- coding: utf-8 -*-
"""Training wrappers with the reference's interface (torchkge/utils/training.py:13-200):
``TrainDataLoader(kg, batch_size, sampling_type, use_cuda)`` yielding dict batches
{'h','t','r','nh','nt'} with the whole graph re-corrupted once per epoch iterator
(:73-77), and ``Trainer(model, criterion, kg_train, n_epochs, batch_size, optimizer,
sampling_type='bern', use_cuda=None).run()`` which calls
``model.normalize_parameters()`` after every epoch (:188).  The step itself is the
engine's: K5 corruption, K1 forward/backward through ``Model.forward``.
The engine has no CPU path, so ``use_cuda`` must be 'all' or 'batch'."""
from tqdm.autonotebook import tqdm

from ..data_structures import SmallKG
from ..sampling import BernoulliNegativeSampler, UniformNegativeSampler
from .data import get_n_batches


class TrainDataLoader:
    def __init__(self, kg, batch_size, sampling_type, use_cuda=None):
        if use_cuda not in ('all', 'batch'):
            raise RuntimeError("torchkge_amd trains on MI355X only: use_cuda must be 'all' or 'batch'")
        assert sampling_type in ('unif', 'bern')
        self.h, self.t, self.r = kg.head_idx, kg.tail_idx, kg.relations
        self.use_cuda = use_cuda
        self.b_size = batch_size
        self.iterator = None
        self.sampler = UniformNegativeSampler(kg) if sampling_type == 'unif' else BernoulliNegativeSampler(kg)
        self.tmp_cuda = True
        if use_cuda == 'all':
            self.h, self.t, self.r = self.h.cuda(), self.t.cuda(), self.r.cuda()

    def __len__(self):
        return get_n_batches(len(self.h), self.b_size)

    def __iter__(self):
        self.iterator = TrainDataLoaderIter(self)
        return self.iterator

    def get_counter_examples(self):
        """SmallKG of the negatives of the current epoch (None before the first)."""
        if self.iterator is None:
            return None
        return SmallKG(self.iterator.nh, self.iterator.nt, self.iterator.r)


class TrainDataLoaderIter:
    def __init__(self, loader):
        self.h, self.t, self.r = loader.h, loader.t, loader.r
        self.nh, self.nt = loader.sampler.corrupt_kg(loader.b_size, loader.tmp_cuda)   # whole KG, once per epoch
        self.nh, self.nt = self.nh.cuda(), self.nt.cuda()
        self.use_cuda = loader.use_cuda
        self.b_size = loader.b_size
        self.n_batches = get_n_batches(len(self.h), self.b_size)
        self.current_batch = 0

    def __next__(self):
        if self.current_batch == self.n_batches:
            raise StopIteration
        i = self.current_batch
        self.current_batch += 1
        sl = slice(i * self.b_size, (i + 1) * self.b_size)
        batch = {'h': self.h[sl], 't': self.t[sl], 'r': self.r[sl], 'nh': self.nh[sl], 'nt': self.nt[sl]}
        if self.use_cuda == 'batch':
            batch = {k: v.cuda() for k, v in batch.items()}
        return batch

    def __iter__(self):
        return self


class Trainer:
    def __init__(self, model, criterion, kg_train, n_epochs, batch_size, optimizer, sampling_type='bern',
                 use_cuda=None):
        self.model = model
        self.criterion = criterion
        self.kg_train = kg_train
        self.use_cuda = use_cuda
        self.n_epochs = n_epochs
        self.optimizer = optimizer
        self.sampling_type = sampling_type
        self.batch_size = batch_size
        self.n_triples = len(kg_train)
        self.counter_examples = None

    def process_batch(self, current_batch):
        self.optimizer.zero_grad()
        h, t, r = current_batch['h'], current_batch['t'], current_batch['r']
        nh, nt = current_batch['nh'], current_batch['nt']
        p, n = self.model(h, t, r, nh, nt)
        loss = self.criterion(p, n)
        loss.backward()
        self.optimizer.step()
        return loss.detach().item()

    def run(self):
        self.model.cuda()
        self.criterion.cuda()
        iterator = tqdm(range(self.n_epochs), unit='epoch')
        data_loader = TrainDataLoader(self.kg_train, batch_size=self.batch_size, sampling_type=self.sampling_type,
                                      use_cuda=self.use_cuda)
        for epoch in iterator:
            sum_ = 0
            for batch in data_loader:
                sum_ += self.process_batch(batch)
            iterator.set_description('Epoch {} | mean loss: {:.5f}'.format(epoch + 1, sum_ / len(data_loader)))
            self.model.normalize_parameters()
            self.counter_examples = data_loader.get_counter_examples()

    def get_counter_examples(self):
        return self.counter_examples
