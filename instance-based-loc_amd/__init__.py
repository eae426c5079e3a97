"""instance-based-loc on MI355X: the embed -> match -> assign -> register hot path of
ObjectMemory.localise() as hand-written HIP (gfx950) behind a C-ABI, with a Python host layer that
mirrors the reference's module surface (utils.embeddings, utils.similarity_volume,
utils.fpfh_register, object_memory.object_memory)."""
__version__ = "0.1.0"
