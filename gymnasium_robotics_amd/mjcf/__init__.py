from .compiler import CompiledModel, MjcfCompiler, compile_mjcf, load_model, save_model, pack_blob, field_list  # noqa: F401
