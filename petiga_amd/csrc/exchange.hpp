// exchange.hpp -- ghost-row pack/unpack for the multi-GPU path (one process per GPU).
extern "C" int IGXGetNeighborCount(IGX g, int *nsend, int *nrecv) { NEEDIGA(g); if (nsend) *nsend = 0; if (nrecv) *nrecv = 0; return 0; }
extern "C" int IGXGetNeighborInfo(IGX g, int, int, int *, int64_t *, int64_t *) { NEEDIGA(g); return fail(IGX_ERR_ARG_OUTOFRANGE, "no such neighbour"); }
extern "C" int IGXPackGhostRows(IGX g, IGXMat, IGXVec, int, double *) { NEEDIGA(g); return fail(IGX_ERR_ARG_OUTOFRANGE, "no such neighbour"); }
extern "C" int IGXUnpackGhostRows(IGX g, IGXMat, IGXVec, int, const double *) { NEEDIGA(g); return fail(IGX_ERR_ARG_OUTOFRANGE, "no such neighbour"); }
