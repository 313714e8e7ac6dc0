"""`graphormer/modelGNN.py:21-74`: GraphConvolution / GCN with a dense normalised adjacency.  These are
plain library GEMMs (hipBLASLt through torch); they produce the POI / category tables the node-feature
gathers read (model_fqandtoyo.py:1236-1237).  Parameter names and init match the reference."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import Parameter


class GraphConvolution(nn.Module):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.weight = Parameter(torch.empty(in_features, out_features))
        if bias:
            self.bias = Parameter(torch.empty(out_features))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1.0 / math.sqrt(self.weight.size(1))
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)

    def forward(self, input, adj):
        # X.W stays fp32 (raw features such as lat/lon need the mantissa); only the big dense adjacency
        # product runs in `adj`'s dtype (fp32, or bf16 in the bf16 configuration)
        with torch.autocast(device_type=input.device.type, enabled=False):
            support = torch.mm(input.float(), self.weight)
            output = torch.mm(adj, support.to(adj.dtype)).float()
            if self.bias is not None:
                output = output + self.bias
        return output


class GCN(nn.Module):
    def __init__(self, ninput, nhid, noutput, dropout):
        super().__init__()
        self.gcn = nn.ModuleList()
        self.dropout = dropout
        self.leaky_relu = nn.LeakyReLU(0.2)
        channels = [ninput] + nhid + [noutput]
        for i in range(len(channels) - 1):
            self.gcn.append(GraphConvolution(channels[i], channels[i + 1]))

    def forward(self, x, adj):
        for i in range(len(self.gcn) - 1):
            x = self.leaky_relu(self.gcn[i](x, adj))
        x = F.dropout(x, self.dropout, training=self.training)
        return self.gcn[-1](x, adj)
