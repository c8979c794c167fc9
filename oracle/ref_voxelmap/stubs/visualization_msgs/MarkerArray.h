// ORACLE / TEST INFRASTRUCTURE ONLY -- empty stand-in (nothing of this header is used by the compiled reference excerpts)
#pragma once
