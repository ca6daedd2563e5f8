"""Drop-in plugin modules: put this directory on sys.path and the reference's own
`import fused`, `import upfirdn2d_op`, `import neural_renderer as nr` bind to libg2s.so
(see INTEGRATION.md)."""
