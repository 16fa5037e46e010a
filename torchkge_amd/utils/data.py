# -*- coding: utf-8 -*-
"""Sequential, un-shuffled batch iterator over a knowledge graph's three index
vectors (reference utils/data.py:63-151)."""


def get_n_batches(n, b_size):
    """ceil(n / b_size)   (data.py:63-80)."""
    return n // b_size + (1 if n % b_size > 0 else 0)


class DataLoader:
    """Contiguous slices of (head_idx, tail_idx, relations).
    use_cuda: None | 'all' (move the KG once) | 'batch' (move each batch)."""

    def __init__(self, kg, batch_size, use_cuda=None):
        self.h, self.t, self.r = kg.head_idx, kg.tail_idx, kg.relations
        self.use_cuda = use_cuda
        self.batch_size = batch_size
        if use_cuda is not None and use_cuda == 'all':
            self.h, self.t, self.r = self.h.cuda(), self.t.cuda(), self.r.cuda()

    def __len__(self):
        return get_n_batches(len(self.h), self.batch_size)

    def __iter__(self):
        b = self.batch_size
        for i in range(len(self)):
            sl = slice(i * b, (i + 1) * b)
            if self.use_cuda is not None and self.use_cuda == 'batch':
                yield self.h[sl].cuda(), self.t[sl].cuda(), self.r[sl].cuda()
            else:
                yield self.h[sl], self.t[sl], self.r[sl]
