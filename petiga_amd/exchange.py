"""Ghost-row reduction between ranks (one process per GPU): transport over torch.distributed (RCCL)."""


class GhostExchange:
    def __init__(self, iga, A, b):
        raise NotImplementedError("multi-GPU ghost-row exchange is not wired yet")
