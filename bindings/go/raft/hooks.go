package raft

// The lines a maintainer adds to src/raft/config.go so that the unmodified tests of test_test.go run
// against the engine.  Kept as comments: config.go is a _test-less file of the reference and is not
// copied here.
//
//   type config struct { ...; grp *Group }
//   make_config, after `cfg.n = n` (config.go:71):      cfg.grp = BeginConfig(n, 4096)
//   cleanup, first line (config.go:224):                 cfg.grp.EndConfig()
//   connect / disconnect, first line (config.go:235,258): cfg.grp.SetConnected(i, true|false)
//   setunreliable / setlongreordering (config.go:288,296): cfg.grp.SetUnreliable(unrel, longrel)
//   rpcTotal (config.go:284):                             return cfg.grp.RPCTotal()
//
// crash1 / start1 (config.go:99-215) need no change: they call Kill, Persister.Copy and Make.
